"""ctypes binding of libmi355_hotpath.so (the C ABI declared in include/mi355_hotpath.h).

The library is the product path: if it is missing or fails to load, importing the ops raises --
there is no CPU or PyTorch fallback behind these functions.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355_hotpath.so")

MI355_OK = 0
F32, BF16, F16 = 0, 1, 2
I32, I64 = 0, 1

P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> (restype, argtypes); mirrors include/mi355_hotpath.h one to one
_SIGNATURES = {
    "mi355_abi_version": (I, []),
    "mi355_last_error": (c_char_p, []),
    "mi355_rmsnorm": (I, [P, P, P, L, L, L, L, F, I, P]),
    "mi355_fused_add_rmsnorm": (I, [P, P, P, L, L, F, I, P]),
    "mi355_silu_and_mul": (I, [P, P, L, L, I, P]),
    "mi355_softcap_inplace": (I, [P, L, F, P]),
    "mi355_rope_inplace": (I, [P, P, P, P, P, I, L, I, I, I, I, L, L, L, I, P, P, P, I, L, I, P]),
    "mi355_set_kv_buffer": (I, [P, P, P, P, P, I, L, L, L, L, L, L, I, P]),
    "mi355_set_kv_buffer_heads": (I, [P, P, P, P, P, I, L, I, I, I, L, L, L, I, P]),
    "mi355_embedding": (I, [P, P, P, L, L, L, L, I, P]),
    "mi355_create_kv_indices": (I, [P, P, I, P, I, P, P, P, L, L, P]),
    "mi355_write_req_to_token": (I, [P, P, P, P, P, P, L, L, P]),
    "mi355_compute_position": (I, [P, P, P, P, L, P]),
    "mi355_prepare_decode": (I, [P, P, P, P, P, L, L, P]),
    "mi355_decode_attention": (I, [P, P, P, P, P, I, P, P, P, L, P, P, P, P, I, L, I, I, I, I, L, L, F, F, I, P]),
    "mi355_decode_num_splits": (I, [L, I, L, I]),
    "mi355_extend_attention": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, L, L, I, I, I, I, L, L, L,
                                   L, F, F, I, I, P]),
    "mi355_linear_workspace_bytes": (L, [L, L, L]),
    "mi355_linear": (I, [P, P, P, P, L, L, L, L, P, L, I, I, P]),
    "mi355_packed_weight_elems": (L, [L, L]),
    "mi355_pack_weight": (I, [P, P, L, L, I, P]),
    "mi355_decode_attention_static": (I, [P, P, P, P, P, I, P, P, P, L, I, L, I, I, I, I, L, L, F, F, I, P]),
    "mi355_decode_attention_mla_fused_rope_supported": (I, [I, I, I, I]),
    "mi355_decode_attention_mla_fused_rope": (I, [P, P, L, P, L, P, P, P, I, I, P, P, P, I, P, P, P, L, P, P, P, P, I, L, I, I, I,
                                                  L, L, F, F, I, P]),
    "mi355_decode_attention_mla": (I, [P, P, P, P, I, P, P, P, L, P, P, P, P, I, L, I, I, I, L, L, F, F, I, P]),
    "mi355_extend_attention_mla_supported": (I, [I, I, I, I]),
    "mi355_extend_attention_mla": (I, [P, P, P, P, P, P, P, L, L, L, I, I, I, L, L, F, F, I, P]),
    "mi355_mla_latent_norms": (I, [P, P, P, P, L, P, I, P, P, P, P, P, P, L, I, I, I, F, I, P]),
    "mi355_mla_absorb": (I, [P, P, P, L, I, I, I, I, L, L, L, L, I, P]),
    "mi355_mla_prefill_assemble": (I, [P, P, P, P, P, L, P, L, P, P, I, I, P, P, L, I, I, I, I, I, L, L, I, P]),
    "mi355_linear_packed": (I, [P, P, P, P, L, L, L, L, P, L, I, I, P]),
    "mi355_linear_packed_silu_mul": (I, [P, P, P, L, L, L, L, P, L, I, P]),
    "mi355_linear_tiled_supported": (I, [L, L, L, I, I]),
    "mi355_linear_tiled": (I, [P, P, P, P, L, L, L, L, I, I, P]),
    "mi355_fp8_scaled_mm_tiled_supported": (I, [L, L, L, I, I]),
    "mi355_fp8_scaled_mm_tiled": (I, [P, P, P, P, P, P, L, L, L, I, I, P]),
    "mi355_qkv_proj_rope_tiled_supported": (I, [L, L, I, I, I, I, I, I, I]),
    "mi355_qkv_proj_rope_tiled": (I, [P, P, P, P, P, L, L, L, I, I, I, P, P, I, I, I, P, P, P, I, L, I, P]),
    "mi355_fp8_packed_workspace_bytes": (L, [L, L, L]),
    "mi355_fp8_scaled_mm_packed": (I, [P, P, P, P, P, P, L, L, L, P, L, I, P]),
    "mi355_prefetch_next": (I, [P, L, L, L, I, L]),
    "mi355_decode_attention_fused_qkv": (I, [P, I, P, P, P, P, P, I, P, P, P, P, I, P, P, P, L, P, P, P, P, I, L, I, I, I, L, F, I, P]),
    "mi355_copy_decode_inputs": (I, [P, P, I, L, P]),
    "mi355_alloc_extend": (I, [P, P, P, P, P, P, L, L, P]),
    "mi355_alloc_decode": (I, [P, P, P, P, P, L, L, P]),
    "mi355_fp8_packed_splits": (I, [L, L, L]),
    "mi355_fp8_packed_partials": (I, [P, P, P, L, L, L, P]),
    "mi355_rmsnorm_ex": (I, [P, P, P, P, P, I, P, P, P, P, L, L, F, I, P]),
    "mi355_silu_and_mul_quant_fp8": (I, [P, P, P, L, L, L, I, P]),
    "mi355_linear_packed_splits": (I, [L, L, L]),
    "mi355_linear_packed_partials": (I, [P, P, P, L, L, L, L, I, P]),
    "mi355_ar_region_bytes": (L, [L]),
    "mi355_ar_alloc_region": (I, [P, L]),
    "mi355_ar_free_region": (I, [P]),
    "mi355_ar_ipc_handle_bytes": (I, []),
    "mi355_ar_get_ipc_handle": (I, [P, P]),
    "mi355_ar_open_ipc_handle": (I, [P, P]),
    "mi355_ar_close_ipc_handle": (I, [P]),
    "mi355_ar_init": (I, [P, P, I, I, L]),
    "mi355_ar_dispose": (I, [P]),
    "mi355_ar_set_timeout_ms": (I, [P, L]),
    "mi355_ar_poll_status": (I, [P]),
    "mi355_ar_choose_algo": (I, [I, L]),
    "mi355_ar_all_reduce": (I, [P, P, P, L, I, I, P]),
    "mi355_ar_all_gather": (I, [P, P, P, L, P]),
    "mi355_ar_fused_add_rmsnorm": (I, [P, P, P, P, L, L, F, I, P]),
    "mi355_ar_status": (I, [P, P]),
    "mi355_fused_add_rmsnorm_partials": (I, [P, P, P, P, I, L, L, F, I, P]),
    "mi355_per_token_quant_fp8": (I, [P, P, P, L, L, L, L, I, P]),
    "mi355_fp8_workspace_bytes": (L, [L, L, L]),
    "mi355_fp8_scaled_mm": (I, [P, P, P, P, P, P, L, L, L, P, L, I, P]),
    "mi355_argmax": (I, [P, P, L, L, L, P]),
    "mi355_argmax_workspace_bytes": (L, [L]),
    "mi355_argmax_chunked": (I, [P, P, L, L, L, P, L, P]),
    "mi355_softmax_temperature": (I, [P, P, L, L, L, P]),
    "mi355_top_k_top_p_min_p_sample": (I, [P, P, P, P, P, P, L, L, L, P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
# extra entry points of the development build (include/mi355_hotpath_dev.h)
_DEV_SIGNATURES = {
    "mi355_debug_set": (None, [I, I]),
    "mi355_debug_read_bw": (I, [P, L, I, I, P, P]),
    "mi355_debug_xcc_map": (I, [P, I, P]),
}
_use_dev = False


def use_dev_library(path: str = None) -> None:
    """Tools only: load libmi355_hotpath_dev.so (ablation switches + bandwidth probes) instead of the
    product library; `path` names another development build (an A/B of two source trees).  Must be
    called before the first load()."""
    global _use_dev, _dev_path
    assert _lib is None, "use_dev_library() must come before the library is loaded"
    _use_dev = True
    _dev_path = path

_dev_path = None

_lib = None


class Mi355Error(RuntimeError):
    pass


def load():
    """Load the shared library once and type every entry point.  Raises if anything the
    header declares is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = (_dev_path or LIB_PATH.replace("libmi355_hotpath.so", "libmi355_hotpath_dev.so")) if _use_dev else LIB_PATH
    if not os.path.exists(path):
        raise Mi355Error(
            f"{path} not found: build it with `python -m sglang_custom_amd.build{' --dev' if _use_dev else ''}` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    # torch (if used by the caller) must be imported first so that its libamdhip64.so.7 is the
    # one HIP runtime in the process; stream handles are only meaningful within one runtime.
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in list(_SIGNATURES.items()) + (list(_DEV_SIGNATURES.items()) if _use_dev else []):
        fn = getattr(lib, name, None)
        if fn is None:
            raise Mi355Error(f"{path} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    if lib.mi355_abi_version() != 1:
        raise Mi355Error("libmi355_hotpath.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != MI355_OK:
        msg = load().mi355_last_error()
        raise Mi355Error(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
