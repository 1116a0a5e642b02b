"""torch.ops.mi355_sgl_kernel.*: the hot-path kernels registered under the reference's op schemas
(sgl-kernel/csrc/common_extension.cc:69-130).  CPU: every schema is registered with the reference's
argument list.  GPU: a call through the dispatcher gives the bits of the direct call through ops.py
(which the other GPU tests hold to the oracle)."""
import pytest
import torch

# the reference's schema strings for the ops registered here (common_extension.cc:69,72,81,91-92,107-108,129)
REFERENCE_SCHEMAS = {
    "rmsnorm": "rmsnorm(Tensor! output, Tensor input, Tensor weight, float eps, bool enable_pdl) -> ()",
    "fused_add_rmsnorm": "fused_add_rmsnorm(Tensor! input, Tensor! residual, Tensor weight, float eps, bool enable_pdl) -> ()",
    "silu_and_mul": "silu_and_mul(Tensor! out, Tensor input, int cuda_stream) -> ()",
    "apply_rope_pos_ids_cos_sin_cache": "apply_rope_pos_ids_cos_sin_cache(Tensor q, Tensor k, Tensor! q_rope, Tensor! k_rope, "
                                        "Tensor cos_sin_cache, Tensor pos_ids, bool interleave, int cuda_stream) -> ()",
    "fp8_scaled_mm": "fp8_scaled_mm(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, ScalarType out_dtype, "
                     "Tensor? bias) -> Tensor",
    "sgl_per_token_quant_fp8": "sgl_per_token_quant_fp8(Tensor input, Tensor output_q, Tensor output_s) -> ()",
}


def _ns():
    from sglang_custom_amd import torch_ops

    return getattr(torch.ops, torch_ops.NAMESPACE), torch_ops


def test_schemas_match_the_reference():
    ns, torch_ops = _ns()
    assert set(torch_ops.OP_NAMES) == set(REFERENCE_SCHEMAS)
    for name, text in REFERENCE_SCHEMAS.items():
        want = torch._C.parse_schema("x::" + text)
        got = getattr(ns, name).default._schema
        assert [(a.name, str(a.type), a.alias_info is not None and a.alias_info.is_write) for a in got.arguments] == \
               [(a.name, str(a.type), a.alias_info is not None and a.alias_info.is_write) for a in want.arguments], name
        assert [str(r.type) for r in got.returns] == [str(r.type) for r in want.returns], name


@pytest.mark.gpu
def test_dispatch_equals_direct_calls():
    ns, _ = _ns()
    from sglang_custom_amd import ops

    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(37, 1024, device="cuda", generator=g).bfloat16()
    w = torch.randn(1024, device="cuda", generator=g).bfloat16()
    out = torch.empty_like(x)
    ns.rmsnorm(out, x, w, 1e-5, False)
    assert torch.equal(out, ops.rmsnorm(x, w, 1e-5))
    res = torch.randn_like(x)
    x1, r1, x2, r2 = x.clone(), res.clone(), x.clone(), res.clone()
    ns.fused_add_rmsnorm(x1, r1, w, 1e-5, False)
    ops.fused_add_rmsnorm(x2, r2, w, 1e-5)
    assert torch.equal(x1, x2) and torch.equal(r1, r2)
    out = torch.empty(37, 512, device="cuda", dtype=torch.bfloat16)
    ns.silu_and_mul(out, x, 0)
    assert torch.equal(out, ops.silu_and_mul(x))
    # rope: [tokens, heads, head_dim], fp32 cos|sin cache, both styles
    q = torch.randn(37, 8, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(37, 2, 128, device="cuda", generator=g).bfloat16()
    cache = torch.randn(64, 128, device="cuda", generator=g)
    pos = torch.randint(0, 64, (37,), device="cuda", generator=g)
    for interleave in (False, True):
        q1, k1 = torch.empty_like(q), torch.empty_like(k)
        ns.apply_rope_pos_ids_cos_sin_cache(q, k, q1, k1, cache, pos, interleave, 0)
        q2, k2 = q.clone(), k.clone()
        ops.rope_inplace(pos, q2.view(37, -1), k2.view(37, -1), 128, cache, not interleave)
        assert torch.equal(q1, q2) and torch.equal(k1, k2)
    # fp8: quantise, then both GEMM sizes (decode-sized -> row-major kernel, prefill-sized -> tile kernel)
    from sglang_custom_amd.layers.quantization.w8a8_fp8 import quantize_weight_per_channel

    for m in (16, 300):
        a = torch.randn(m, 512, device="cuda", generator=g).bfloat16()
        wq, ws = quantize_weight_per_channel((torch.randn(256, 512, device="cuda", generator=g) * 0.05).bfloat16())
        aq = torch.empty(m, 512, device="cuda", dtype=torch.float8_e4m3fn)
        asc = torch.empty(m, 1, device="cuda", dtype=torch.float32)
        ns.sgl_per_token_quant_fp8(a, aq, asc)
        xq, xs = ops.per_token_quant_fp8(a)
        assert torch.equal(aq.view(torch.uint8), xq.view(torch.uint8)) and torch.equal(asc, xs)
        y = ns.fp8_scaled_mm(aq, wq.t(), asc, ws, torch.bfloat16, None)
        direct = (ops.fp8_scaled_mm_tiled if m > 64 else ops.fp8_scaled_mm)(xq, wq, xs.view(-1), ws.view(-1).contiguous(),
                                                                          torch.bfloat16, None)
        assert torch.equal(y, direct)
