"""The prefill-sized projection kernel (mi355_linear_tiled, through the C ABI) -- the reference's
UnquantizedLinearMethod.apply = F.linear (python/sglang/srt/layers/quantization/unquant.py:110-122)
and LlamaMLP's gate_up_proj + SiluAndMul (models/llama.py:94-98, layers/activation.py:55-65).

Small shapes are compared with F.linear on the CPU (the reference's own call) and the oracle's
silu_and_mul; at the benchmark's shapes the oracle would take minutes, so parity is checked through
size-independent properties (exact power-of-two scaling, a row's bits do not depend on the rows that
share the launch, permuting weight rows permutes outputs, fused SiLU-and-mul == the unfused pair bit
for bit) plus float64 evaluations of sampled rows on the GPU.  Tolerance: one rounding of the output
dtype on top of fp32 accumulation noise, 2^-7 * max|ref| + 1e-3 for bf16 (8 significand bits)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from sglang_custom_amd import ops as _ops
    from sglang_custom_amd import _lib

    _lib.load()
    return _ops


def _data(m, n, k, dtype, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(m, k, device=DEV, generator=g).to(dtype)
    w = (torch.randn(n, k, device=DEV, generator=g) * k ** -0.5).to(dtype)
    return x, w


# ragged edges in M and N, the smallest K, K = 64 * odd, one tile, many tiles
SMALL = [(1, 16, 128), (65, 48, 192), (256, 256, 128), (300, 272, 320), (513, 1040, 256), (100, 4096, 1024)]


@pytest.mark.parametrize("m,n,k", SMALL)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_tiled_vs_oracle(ops, m, n, k, dtype):
    import torch.nn.functional as F   # F.linear on CPU IS the reference's apply (unquant.py:122)

    x, w = _data(m, n, k, dtype, m + n + k)
    bias = torch.randn(n, device=DEV).to(dtype)
    for b in (None, bias):
        y = ops.linear_tiled(x, w, b)
        ref = F.linear(x.cpu(), w.cpu(), None if b is None else b.cpu())
        # the oracle accumulates in another order: allow one unit in the last place of the output dtype
        tol = (2 ** -7 if dtype == torch.bfloat16 else 2 ** -10) * ref.float().abs().max().item() + 1e-3
        assert (y.cpu().float() - ref.float()).abs().max().item() <= tol
        # and float64 on the GPU: the kernel's fp32 accumulation, rounded once
        r64 = x.double() @ w.double().t() + (0 if b is None else b.double())
        assert (y.double() - r64).abs().max().item() <= tol


@pytest.mark.parametrize("m,n,k", [(1, 32, 128), (300, 544, 192), (777, 2048, 512)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_tiled_silu_mul_equals_unfused(ops, m, n, k, dtype):
    """SiLU-and-mul in the epilogue rounds gate and up to the model dtype first: the same bits as
    the kernel's plain output followed by mi355_silu_and_mul, and the oracle's values."""
    import torch.nn.functional as F

    from oracle import ops_ref

    x, w = _data(m, n, k, dtype, 7 * m + n)
    fused = ops.linear_tiled(x, w, None, silu_mul=True)
    assert fused.shape == (m, n // 2)
    assert torch.equal(fused, ops.silu_and_mul(ops.linear_tiled(x, w)))
    ref = ops_ref.silu_and_mul(F.linear(x.cpu(), w.cpu()))
    tol = (2 ** -6 if dtype == torch.bfloat16 else 2 ** -9) * ref.float().abs().max().item() + 1e-3
    assert (fused.cpu().float() - ref.float()).abs().max().item() <= tol


def test_linear_tiled_refuses_what_it_does_not_serve(ops):
    from sglang_custom_amd._lib import Mi355Error

    x, w = _data(8, 24, 128, torch.bfloat16, 0)   # N % 16 != 0
    assert not ops.linear_tiled_supported(8, 24, 128, torch.bfloat16)
    with pytest.raises(Mi355Error):
        ops.linear_tiled(x, w)
    x, w = _data(8, 32, 96, torch.bfloat16, 0)    # K % 64 != 0
    assert not ops.linear_tiled_supported(8, 32, 96, torch.bfloat16)
    with pytest.raises(Mi355Error):
        ops.linear_tiled(x, w)
    assert not ops.linear_tiled_supported(8, 32, 128, torch.float32)


# Llama-3-8B projections (qkv, o, gate_up, down) and the Llama-3-70B TP=8 shard shapes
FULL = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (1280, 8192), (8192, 1024), (7168, 8192), (8192, 3584)]


@pytest.mark.parametrize("n,k", FULL)
def test_linear_tiled_properties_at_full_size(ops, n, k):
    m = 4096 + 77
    x, w = _data(m, n, k, torch.bfloat16, n + k)
    y = ops.linear_tiled(x, w)
    # exact under power-of-two scaling of the activations
    assert torch.equal(ops.linear_tiled(x * 2, w), y * 2)
    assert torch.equal(ops.linear_tiled(x * 0.25, w), y * 0.25)
    # a row does not see its neighbours: same bits alone, in a ragged batch, at another tile position
    for rows in ([0], [m - 1], list(range(300, 300 + 65)), list(range(m - 1, m - 600, -1))):
        idx = torch.tensor(rows, device=DEV)
        assert torch.equal(ops.linear_tiled(x[idx].contiguous(), w), y[idx]), rows[:2]
    # a strided activation (a column slice of a wider buffer) reads the same values
    wide = torch.zeros(m, k + 64, device=DEV, dtype=x.dtype)
    wide[:, :k] = x
    assert torch.equal(ops.linear_tiled(wide[:, :k], w), y)
    # permuting output channels permutes the outputs
    perm = torch.randperm(n, device=DEV)
    assert torch.equal(ops.linear_tiled(x, w[perm].contiguous()), y[:, perm])
    # float64 evaluation of sampled rows
    rows = torch.randint(0, m, (48,), device=DEV)
    ref = x[rows].double() @ w.double().t()
    assert (y[rows].double() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-3
    # the gate_up shapes: fused SiLU-and-mul == unfused, bit for bit
    if n in (28672, 7168):
        assert torch.equal(ops.linear_tiled(x, w, None, silu_mul=True), ops.silu_and_mul(y))


def test_prefill_through_the_tiled_gemm_matches_the_library_path(ops):
    """A prefill of the bf16 model with PREFILL_GEMM = "tiled" against the same prefill through the
    library GEMM: logits within bf16 accumulation-order noise, and the greedy tokens agree wherever
    the library path's top-2 margin exceeds that noise."""
    import numpy as np

    from sglang_custom_amd import bench_one_batch as bob
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.layers.quantization import unquant
    from sglang_custom_amd.managers.schedule_batch import Req
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.server_args import ServerArgs

    cfg = hf_config_dict("tiny-llama-d128", "bfloat16")
    w = make_synthetic_llama_weights(cfg, seed=4321, dtype=torch.bfloat16)
    rng = np.random.RandomState(3)
    prompts = [rng.randint(0, cfg["vocab_size"], n).tolist() for n in (700, 513, 300, 64)]
    sa = ServerArgs(dtype="bfloat16", max_total_tokens=4096, max_running_requests=8, disable_cuda_graph=True)
    mr = ModelRunner(ModelConfig(cfg, dtype="bfloat16"), sa, weights={k: v.cuda() for k, v in w.items()})
    outs = {}
    keep, keep_min = unquant.PREFILL_GEMM, unquant.TILED_MIN_TILES
    try:
        for mode in ("library", "tiled"):
            unquant.PREFILL_GEMM, unquant.TILED_MIN_TILES = mode, 1
            mr.token_to_kv_pool_allocator.clear()
            mr.req_to_token_pool.clear()
            nxt, logits, _ = bob.extend([Req(i, p) for i, p in enumerate(prompts)], mr)
            outs[mode] = (nxt.cpu(), logits.float().cpu())
    finally:
        unquant.PREFILL_GEMM, unquant.TILED_MIN_TILES = keep, keep_min
    (t_lib, l_lib), (t_til, l_til) = outs["library"], outs["tiled"]
    assert (l_lib - l_til).abs().max().item() < 5e-2
    top2 = l_lib.topk(2, dim=-1).values
    sure = (top2[:, 0] - top2[:, 1]) > 1e-1
    assert torch.equal(t_lib[sure], t_til[sure])


# ---------------------------------------------------------------------------------- fp8 W8A8
def _fp8_data(ops, m, n, k, seed, dtype=torch.bfloat16, with_x=False):
    from sglang_custom_amd.layers.quantization.w8a8_fp8 import quantize_weight_per_channel

    x, w = _data(m, n, k, dtype, seed)
    xq, xs = ops.per_token_quant_fp8(x)
    wq, ws = quantize_weight_per_channel(w)
    if with_x:
        return x, xq, xs, wq, ws.contiguous()
    return xq, xs, wq, ws.contiguous()


@pytest.mark.parametrize("m,n,k", [(1, 16, 256), (65, 48, 384), (300, 272, 256), (513, 1040, 512), (100, 4096, 1024)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fp8_tiled_vs_oracle_and_the_row_major_kernel(ops, m, n, k, dtype):
    """apply_fp8_linear (fp8_utils.py:676-749) at prefill sizes: against the oracle's fp8_linear on the
    CPU (per-token quantisation, fp32 accumulation of the fp8 products, * x_scale * w_scale, one
    rounding) and against mi355_fp8_scaled_mm, the library's row-major kernel for the same contract.
    Tolerance: one rounding of the output dtype (the accumulation orders differ)."""
    from oracle import ops_ref

    x, xq, xs, wq, ws = _fp8_data(ops, m, n, k, m * 3 + n + k, dtype, with_x=True)
    bias = torch.randn(n, device=DEV).to(dtype)
    for b in (None, bias):
        y = ops.fp8_scaled_mm_tiled(xq, wq, xs, ws, dtype, b)
        ref = ops_ref.fp8_linear(x.cpu(), wq.cpu(), ws.cpu(), None if b is None else b.cpu().float())
        tol = (2 ** -7 if dtype == torch.bfloat16 else 2 ** -10) * ref.float().abs().max().item() + 1e-3
        assert (y.cpu().float() - ref.float()).abs().max().item() <= tol
        other = ops.fp8_scaled_mm(xq, wq, xs, ws, dtype, b)
        assert (y.float() - other.float()).abs().max().item() <= tol


@pytest.mark.parametrize("m,n,k", [(300, 544, 256), (777, 2048, 512)])
def test_fp8_tiled_silu_mul_equals_unfused(ops, m, n, k):
    xq, xs, wq, ws = _fp8_data(ops, m, n, k, m + n)
    fused = ops.fp8_scaled_mm_tiled(xq, wq, xs, ws, torch.bfloat16, None, silu_mul=True)
    assert torch.equal(fused, ops.silu_and_mul(ops.fp8_scaled_mm_tiled(xq, wq, xs, ws, torch.bfloat16)))


@pytest.mark.parametrize("n,k", [(6144, 4096), (28672, 4096), (4096, 14336), (7168, 8192)])
def test_fp8_tiled_properties_at_full_size(ops, n, k):
    m = 2048 + 77
    xq, xs, wq, ws = _fp8_data(ops, m, n, k, n + k)
    y = ops.fp8_scaled_mm_tiled(xq, wq, xs, ws, torch.bfloat16)
    # power-of-two scaling of either scale vector is exact
    assert torch.equal(ops.fp8_scaled_mm_tiled(xq, wq, xs * 2, ws, torch.bfloat16), y * 2)
    assert torch.equal(ops.fp8_scaled_mm_tiled(xq, wq, xs, ws * 0.25, torch.bfloat16), y * 0.25)
    # row independence and weight-row permutation
    for rows in ([0], [m - 1], list(range(300, 365))):
        idx = torch.tensor(rows, device=DEV)
        assert torch.equal(ops.fp8_scaled_mm_tiled(xq[idx].contiguous(), wq, xs[idx].contiguous(), ws, torch.bfloat16),
                           y[idx])
    perm = torch.randperm(n, device=DEV)
    assert torch.equal(ops.fp8_scaled_mm_tiled(xq, wq[perm].contiguous(), xs, ws[perm].contiguous(), torch.bfloat16),
                       y[:, perm])
    # float64 evaluation of sampled rows: the fp8 products are exact, so this is the same formula
    rows = torch.randint(0, m, (32,), device=DEV)
    ref = (xq[rows].double() @ wq.double().t()) * xs[rows].double().view(-1, 1) * ws.double().view(1, -1)
    assert (y[rows].double() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-3
    if n in (28672, 7168):
        assert torch.equal(ops.fp8_scaled_mm_tiled(xq, wq, xs, ws, torch.bfloat16, None, silu_mul=True),
                           ops.silu_and_mul(y))


def test_fp8_prefill_mlp_through_the_fused_gate_up(ops):
    """LlamaMLP on fp8 W8A8 weights at a prefill size: the fused gate_up launch (quantise -> GEMM with
    scales and SiLU-and-mul in the epilogue) gives the bits of the unfused sequence on the same kernel."""
    from sglang_custom_amd.layers.quantization import w8a8_fp8
    from sglang_custom_amd.models.llama import LlamaMLP

    torch.manual_seed(0)
    mlp = LlamaMLP(512, 1536, torch.bfloat16, DEV, quant_method_factory=w8a8_fp8.W8A8Fp8LinearMethod)
    for lin in (mlp.gate_up_proj, mlp.down_proj):
        lin.weight.data.normal_(0, 0.03)
        lin.quant_method.process_weights_after_loading(lin)
    x = torch.randn(700, 512, device=DEV).bfloat16()
    keep = (w8a8_fp8.PREFILL_GEMM, w8a8_fp8.TILED_MIN_TILES)
    try:
        w8a8_fp8.TILED_MIN_TILES = 1
        w8a8_fp8.PREFILL_GEMM = "auto"
        fused = mlp(x)
        w8a8_fp8.PREFILL_GEMM = "tiled"
        assert torch.equal(mlp(x), fused)
        w8a8_fp8.PREFILL_GEMM = "library"
        lib = mlp(x)
    finally:
        w8a8_fp8.PREFILL_GEMM, w8a8_fp8.TILED_MIN_TILES = keep
    assert fused.shape == lib.shape == (700, 512)
    assert (fused.float() - lib.float()).abs().max().item() <= 2 ** -6 * lib.float().abs().max().item() + 1e-3


# ------------------------------------------------------- qkv projection + RoPE + KV-cache write
def _qkv_case(m, k, hq, hkv, dtype, seed, cache_f32, loc_dtype=torch.int64, slots=None):
    g = torch.Generator(device=DEV).manual_seed(seed)
    n = (hq + 2 * hkv) * 128
    x = torch.randn(m, k, device=DEV, generator=g).to(dtype)
    w = (torch.randn(n, k, device=DEV, generator=g) * k ** -0.5).to(dtype)
    max_pos = 4096
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, device=DEV).float() / 128))
    fr = torch.arange(max_pos, device=DEV).float()[:, None] * inv[None]
    cs = torch.cat([fr.cos(), fr.sin()], dim=-1)
    cs = cs if cache_f32 else cs.to(dtype)
    pos = torch.randint(0, max_pos, (m,), device=DEV, generator=g)
    slots = slots or m + 37
    loc = torch.randperm(slots, device=DEV, generator=g)[:m].to(loc_dtype)
    return x, w, cs.contiguous(), pos, loc, slots


@pytest.mark.parametrize("m,k,hq,hkv", [(1, 128, 2, 1), (300, 192, 4, 1), (513, 512, 6, 2), (1000, 1024, 32, 8),
                                        (257, 256, 5, 1)])
@pytest.mark.parametrize("dtype,cache_f32", [(torch.bfloat16, True), (torch.bfloat16, False), (torch.float16, True)])
def test_qkv_proj_rope_tiled_equals_the_unfused_sequence(ops, m, k, hq, hkv, dtype, cache_f32):
    """LlamaAttention.forward's qkv_proj -> rotary_emb -> set_kv_buffer (models/llama.py:187-199) as one
    launch gives the bits of mi355_linear_tiled followed by mi355_rope_inplace with its cache write: the
    qkv tensor, and every slot of both cache buffers (slots no token maps to keep their old contents)."""
    if (hq + 2 * hkv) % 2:
        assert not ops.qkv_proj_rope_tiled_supported(m, k, hq, hkv, 128, 128, True, dtype)
        return
    x, w, cs, pos, loc, slots = _qkv_case(m, k, hq, hkv, dtype, m + k + hq, cache_f32,
                                          torch.int32 if m % 2 else torch.int64)
    assert ops.qkv_proj_rope_tiled_supported(m, k, hq, hkv, 128, 128, True, dtype)
    kb = torch.full((slots, hkv, 128), 7.0, device=DEV, dtype=dtype)
    vb = torch.full((slots, hkv, 128), -3.0, device=DEV, dtype=dtype)
    kb2, vb2 = kb.clone(), vb.clone()
    qkv = ops.qkv_proj_rope_tiled(x, w, hq, hkv, 128, pos, cs, True, kb, vb, loc)
    ref = ops.linear_tiled(x, w)
    q, kk, v = ref.split([hq * 128, hkv * 128, hkv * 128], dim=-1)
    ops.rope_inplace(pos, q, kk, 128, cs, True, v=v, k_buffer=kb2, v_buffer=vb2, loc=loc)
    assert torch.equal(qkv, ref)
    assert torch.equal(kb, kb2) and torch.equal(vb, vb2)
    # and the cache rows are the rotated k / plain v rows of the result
    assert torch.equal(kb[loc.long()].view(m, -1), qkv[:, hq * 128:(hq + hkv) * 128])
    assert torch.equal(vb[loc.long()].view(m, -1), qkv[:, (hq + hkv) * 128:])


def test_qkv_proj_rope_tiled_vs_oracle(ops):
    """Against the CPU oracle: F.linear, then the oracle's neox rotation (rounding at the dtype like
    apply_rope_with_cos_sin_cache_inplace), within one rounding of the GEMM's accumulation order."""
    import torch.nn.functional as F

    from oracle import ops_ref

    m, k, hq, hkv, dtype = 200, 256, 4, 2, torch.bfloat16
    x, w, cs, pos, loc, slots = _qkv_case(m, k, hq, hkv, dtype, 99, True)
    kb = torch.zeros((slots, hkv, 128), device=DEV, dtype=dtype)
    vb = torch.zeros_like(kb)
    qkv = ops.qkv_proj_rope_tiled(x, w, hq, hkv, 128, pos, cs, True, kb, vb, loc).cpu()
    ref = F.linear(x.cpu(), w.cpu())
    q, kk, v = ref.split([hq * 128, hkv * 128, hkv * 128], dim=-1)
    q, kk = ops_ref.apply_rope(pos.cpu(), q.contiguous(), kk.contiguous(), 128, cs.cpu(), True)
    want = torch.cat([q, kk, v], dim=-1).float()
    assert (qkv.float() - want).abs().max().item() <= 2 ** -6 * want.abs().max().item() + 1e-3


@pytest.mark.parametrize("m,k,hq,hkv", [(300, 256, 4, 1), (1000, 1024, 32, 8)])
def test_qkv_proj_rope_tiled_fp8_equals_the_unfused_sequence(ops, m, k, hq, hkv):
    from sglang_custom_amd.layers.quantization.w8a8_fp8 import quantize_weight_per_channel

    dtype = torch.bfloat16
    x, w, cs, pos, loc, slots = _qkv_case(m, k, hq, hkv, dtype, m + k, True)
    xq, xs = ops.per_token_quant_fp8(x)
    wq, ws = quantize_weight_per_channel(w)
    ws = ws.contiguous().view(-1)
    kb = torch.zeros((slots, hkv, 128), device=DEV, dtype=dtype)
    vb = torch.zeros_like(kb)
    kb2, vb2 = kb.clone(), vb.clone()
    qkv = ops.qkv_proj_rope_tiled(xq, wq, hq, hkv, 128, pos, cs, True, kb, vb, loc, x_scale=xs, w_scale=ws,
                                  out_dtype=dtype)
    ref = ops.fp8_scaled_mm_tiled(xq, wq, xs, ws, dtype)
    q, kk, v = ref.split([hq * 128, hkv * 128, hkv * 128], dim=-1)
    ops.rope_inplace(pos, q, kk, 128, cs, True, v=v, k_buffer=kb2, v_buffer=vb2, loc=loc)
    assert torch.equal(qkv, ref) and torch.equal(kb, kb2) and torch.equal(vb, vb2)


def test_qkv_proj_rope_tiled_refuses_what_it_does_not_serve(ops):
    from sglang_custom_amd._lib import Mi355Error

    assert not ops.qkv_proj_rope_tiled_supported(512, 4096, 32, 8, 64, 64, True, torch.bfloat16)     # head_dim
    assert not ops.qkv_proj_rope_tiled_supported(512, 4096, 32, 8, 128, 64, True, torch.bfloat16)    # partial rotary
    assert not ops.qkv_proj_rope_tiled_supported(512, 4096, 32, 8, 128, 128, False, torch.bfloat16)  # GPT-J style
    assert not ops.qkv_proj_rope_tiled_supported(512, 4096, 3, 1, 128, 128, True, torch.bfloat16)    # odd head count
    assert not ops.qkv_proj_rope_tiled_supported(512, 4096, 32, 8, 128, 128, True, torch.float32)
    x, w, cs, pos, loc, slots = _qkv_case(64, 128, 2, 1, torch.bfloat16, 5, True)
    kb = torch.zeros((slots, 1, 128), device=DEV, dtype=torch.bfloat16)
    with pytest.raises(Mi355Error):
        ops.qkv_proj_rope_tiled(x, w, 2, 1, 128, pos, cs[:, :64].contiguous(), True, kb, kb.clone(), loc)


def test_prefill_through_the_fused_qkv_rope_matches_the_unfused_path(ops):
    """A prefill (bf16 and fp8 W8A8 models) with the fused qkv + RoPE + cache-write launch against the same
    prefill through the separate kernels: bit-identical logits and KV-cache contents."""
    import numpy as np

    from sglang_custom_amd import bench_one_batch as bob
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.layers.quantization import unquant, w8a8_fp8
    from sglang_custom_amd.managers.schedule_batch import Req
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.models import llama
    from sglang_custom_amd.server_args import ServerArgs

    cfg = hf_config_dict("tiny-llama-d128", "bfloat16")
    w = make_synthetic_llama_weights(cfg, seed=4321, dtype=torch.bfloat16)
    rng = np.random.RandomState(3)
    prompts = [rng.randint(0, cfg["vocab_size"], n).tolist() for n in (700, 513, 300, 64)]
    keep = (llama.PREFILL_QKV_ROPE, llama.PREFILL_QKV_MIN_TILES, unquant.PREFILL_GEMM, unquant.TILED_MIN_TILES)
    keep8 = (w8a8_fp8.PREFILL_GEMM, w8a8_fp8.TILED_MIN_TILES)
    try:
        unquant.PREFILL_GEMM, unquant.TILED_MIN_TILES = "tiled", 1   # both legs multiply on the same kernel
        w8a8_fp8.PREFILL_GEMM, w8a8_fp8.TILED_MIN_TILES = "tiled", 1
        llama.PREFILL_QKV_MIN_TILES = 1
        for quant in (None, "w8a8_fp8"):
            sa = ServerArgs(dtype="bfloat16", max_total_tokens=4096, max_running_requests=8,
                            disable_cuda_graph=True, quantization=quant)
            mr = ModelRunner(ModelConfig(cfg, dtype="bfloat16"), sa, weights={k: v.cuda() for k, v in w.items()})
            outs = {}
            for mode in ("off", "auto"):
                llama.PREFILL_QKV_ROPE = mode
                mr.token_to_kv_pool_allocator.clear()
                mr.req_to_token_pool.clear()
                nxt, logits, _ = bob.extend([Req(i, p) for i, p in enumerate(prompts)], mr)
                kv = [t.clone() for t in mr.token_to_kv_pool.get_kv_buffer(0)]
                outs[mode] = (nxt.cpu(), logits.float().cpu(), kv)
            assert torch.equal(outs["off"][1], outs["auto"][1])
            assert torch.equal(outs["off"][0], outs["auto"][0])
            assert all(torch.equal(a, b) for a, b in zip(outs["off"][2], outs["auto"][2]))
    finally:
        llama.PREFILL_QKV_ROPE, llama.PREFILL_QKV_MIN_TILES, unquant.PREFILL_GEMM, unquant.TILED_MIN_TILES = keep
        w8a8_fp8.PREFILL_GEMM, w8a8_fp8.TILED_MIN_TILES = keep8
