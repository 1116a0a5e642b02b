"""torch.ops registration of the hot-path kernels under the reference's op schemas.

The reference exposes its kernels as `torch.ops.sgl_kernel.*` (TORCH_LIBRARY_EXPAND(sgl_kernel, m) in
sgl-kernel/csrc/torch_extension_rocm.cc:21 and common_extension.cc:20); its Python layers call those
ops.  Importing this module registers the same schemas -- argument order, in-place annotations and
return conventions of common_extension.cc:69-130 -- in the namespace `mi355_sgl_kernel`, implemented
for the CUDA (HIP) dispatch key by calls through the C ABI, so a caller written against
`torch.ops.sgl_kernel.<op>` runs on this library by switching the namespace.  (The all-reduce op set
of torch_extension_rocm.cc:25-55 lives in distributed/device_communicators/custom_all_reduce_ops.py:
those ops take process-local handles rather than tensors.)

Only dispatch is added here: every implementation is one call into ops.py."""
import torch

from . import ops

NAMESPACE = "mi355_sgl_kernel"
_library = torch.library.Library(NAMESPACE, "DEF")


def _rmsnorm(output, input, weight, eps, enable_pdl):   # common_extension.cc:69
    ops.rmsnorm(input, weight, eps, out=output)


def _fused_add_rmsnorm(input, residual, weight, eps, enable_pdl):   # common_extension.cc:72
    ops.fused_add_rmsnorm(input, residual, weight, eps)


def _silu_and_mul(out, input, cuda_stream):   # common_extension.cc:81 (the stream argument is the current stream)
    ops.silu_and_mul(input, out=out)


def _apply_rope_pos_ids_cos_sin_cache(q, k, q_rope, k_rope, cos_sin_cache, pos_ids, interleave, cuda_stream):
    """common_extension.cc:90-93; q / k are [tokens, heads, head_dim]; interleave = not is_neox
    (sgl-kernel/python/sgl_kernel/elementwise.py: apply_rope_with_cos_sin_cache_inplace)."""
    if q_rope.data_ptr() != q.data_ptr():
        q_rope.copy_(q)
    if k_rope.data_ptr() != k.data_ptr():
        k_rope.copy_(k)
    t = q_rope.shape[0]
    ops.rope_inplace(pos_ids.to(torch.int64).contiguous(), q_rope.view(t, -1), k_rope.view(t, -1),
                     q_rope.shape[-1], cos_sin_cache, not interleave)


def _fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias):
    """common_extension.cc:107-109: mat_a [M, K] row-major, mat_b [K, N] column-major (= the
    [N, K] row-major weight transposed, as W8A8Fp8LinearMethod stores it, w8a8_fp8.py:115)."""
    wq = mat_b.t()
    assert wq.is_contiguous(), "fp8_scaled_mm: mat_b must be column-major"
    m, n, k = mat_a.shape[0], wq.shape[0], wq.shape[1]
    xs, ws = scales_a.reshape(-1).contiguous(), scales_b.reshape(-1).contiguous()
    if m > 64 and ops.fp8_scaled_mm_tiled_supported(m, n, k, out_dtype):
        return ops.fp8_scaled_mm_tiled(mat_a, wq, xs, ws, out_dtype, bias)
    return ops.fp8_scaled_mm(mat_a, wq, xs, ws, out_dtype, bias)


def _sgl_per_token_quant_fp8(input, output_q, output_s):   # common_extension.cc:129
    xq, xs = ops.per_token_quant_fp8(input)
    output_q.copy_(xq)
    output_s.view(-1).copy_(xs.view(-1))


_OPS = [
    ("rmsnorm(Tensor! output, Tensor input, Tensor weight, float eps, bool enable_pdl) -> ()", _rmsnorm),
    ("fused_add_rmsnorm(Tensor! input, Tensor! residual, Tensor weight, float eps, bool enable_pdl) -> ()",
     _fused_add_rmsnorm),
    ("silu_and_mul(Tensor! out, Tensor input, int cuda_stream) -> ()", _silu_and_mul),
    ("apply_rope_pos_ids_cos_sin_cache(Tensor q, Tensor k, Tensor! q_rope, Tensor! k_rope, Tensor cos_sin_cache, "
     "Tensor pos_ids, bool interleave, int cuda_stream) -> ()", _apply_rope_pos_ids_cos_sin_cache),
    ("fp8_scaled_mm(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, ScalarType out_dtype, Tensor? "
     "bias) -> Tensor", _fp8_scaled_mm),
    ("sgl_per_token_quant_fp8(Tensor input, Tensor output_q, Tensor output_s) -> ()", _sgl_per_token_quant_fp8),
]
OP_NAMES = tuple(schema.split("(")[0] for schema, _ in _OPS)
for _schema, _fn in _OPS:
    _library.define(_schema)
    _library.impl(_schema.split("(")[0], _fn, "CUDA")
