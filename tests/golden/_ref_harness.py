"""DEV-ONLY harness that makes the reference's Python hot path importable in the build
container (CPU only, no network), so golden vectors can be generated from the reference
itself.  Nothing in here ships reference code: it only arranges sys.path / stub modules for
third-party packages the reference imports but the hot path never calls.

Never imported by the product, by `-m gpu` tests, by smoke() or by bench.py: the reference
tree does not exist on the GPU box.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF = os.environ.get("SGLANG_REFERENCE_ROOT", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# Third-party modules the reference imports at module scope that are absent here and are
# not exercised by the CPU hot path.  `vllm` is deliberately NOT stubbed: its ImportError
# is what makes the reference fall back to its own in-tree layers.
_STUBS = [
    "pybase64", "zmq", "uvloop", "orjson", "msgspec", "setproctitle", "IPython",
    "partial_json_parser", "interegular", "compressed_tensors", "torchao", "decord",
    "pynvml", "llguidance", "xgrammar", "outlines", "sgl_kernel", "flashinfer", "aiter",
    "torch_memory_saver", "deep_gemm", "hf_transfer", "python_multipart", "torchvision",
]


class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return name


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Meta(name, (), {"__init__": lambda self, *a, **k: None,
                                "__call__": lambda self, *a, **k: None,
                                # class Foo(stub.Struct, array_like=True, ...) must be definable
                                "__init_subclass__": classmethod(lambda cls, **k: None)})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Serve a permissive stub only when the DIRECT importer is a reference source file."""

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] not in _STUBS:
            return None
        f = sys._getframe(1)
        while f is not None:
            fn = f.f_code.co_filename
            if "importlib" in fn or fn.startswith("<frozen"):
                f = f.f_back
                continue
            if fn.startswith(REF):
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install():
    """Returns after `import sglang.srt...` works.  Requires oracle/_ref/common_ops.so
    (python oracle/build_ref.py) so that the reference keeps its own RMSNorm/SiluAndMul/
    RotaryEmbedding classes (python/sglang/srt/utils.py:203-214 cpu_has_amx_support)."""
    os.environ["SGLANG_USE_CPU_ENGINE"] = "1"
    import torch

    torch.version.hip = None  # the reference's is_hip() (utils.py:97-98) must be False on CPU
    so = os.path.join(REPO, "oracle", "_ref", "common_ops.so")
    if not os.path.exists(so):
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import build_ref

        so = build_ref.build()
    torch.ops.load_library(so)
    sys.path.insert(0, os.path.join(REF, "python"))
    sys.meta_path.insert(0, _Finder())
    cu = types.ModuleType("sglang.srt.configs.utils")
    cu.register_image_processor = lambda *a, **k: None
    cu.register_processor = lambda *a, **k: None
    sys.modules["sglang.srt.configs.utils"] = cu
    import sglang.srt.managers.schedule_batch  # noqa: F401  (first: avoids a circular import)


def use_amx_kernels(flag: bool):
    """Select which of the reference's two CPU paths runs.  False = pure torch-native
    (CustomOp.forward_native, F.linear): the canonical oracle path, works in fp32 and bf16.
    True = the sgl-kernel AMX/AVX512 C++ ops (bf16/fp16 only).  The reference reads the
    switch from module-level constants evaluated at import, so flip those."""
    import importlib

    # rotary_embedding keeps its flag True: its constructor imports vllm otherwise
    # (layers/rotary_embedding.py:103-108); which forward runs is decided by custom_op.
    for name in [
        "sglang.srt.custom_op", "sglang.srt.layers.quantization.unquant",
        "sglang.srt.layers.vocab_parallel_embedding",
        "sglang.srt.layers.activation", "sglang.srt.layers.layernorm",
    ]:
        mod = importlib.import_module(name)
        setattr(mod, "_is_cpu_amx_available", bool(flag))
