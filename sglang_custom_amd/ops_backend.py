"""Which object provides the hot-path ops to the host-side layers.

Product: `sglang_custom_amd.ops` (HIP kernels through the C ABI) -- loading fails loudly when
libmi355_hotpath.so is missing; there is no CPU fallback in this package.  Tests of pure host
logic (batch plumbing, TP sharding over gloo) may install a CPU checker implementation with
set_ops_backend(); that hook exists for tests only and nothing in the package calls it.
"""
_backend = None


def get_ops():
    global _backend
    if _backend is None:
        from . import _lib, ops

        _lib.load()  # raises Mi355Error if the HIP library is absent
        _backend = ops
    return _backend


def set_ops_backend(backend) -> None:
    global _backend
    _backend = backend
