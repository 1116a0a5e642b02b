"""DeepSeek multi-head latent attention as one module on the GPU: prefill through the multi-head
path (latent expanded by kv_b_proj, latent rows written to the compressed cache), then decode steps
through the absorbed path (multi-query attention over the 576-wide cache rows on the matrix cores),
checked at every position against the oracle's float64 causal attention without absorption."""
from types import SimpleNamespace

import pytest
import torch

import cases as _CASES

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _weights(hidden, H, nope, rope, vd, qr, kr, seed):
    g = torch.Generator().manual_seed(seed)

    def r(*shape, scale):
        return torch.randn(*shape, generator=g) * scale

    w = {"kv_a_layernorm": 1 + 0.1 * r(kr, scale=1.0), "kv_b_proj": r(H * (nope + vd), kr, scale=kr ** -0.5),
         "o_proj": r(hidden, H * vd, scale=(H * vd) ** -0.5)}
    if qr is not None:
        w["fused_qkv_a_proj_with_mqa"] = r(qr + kr + rope, hidden, scale=hidden ** -0.5)
        w["q_a_layernorm"] = 1 + 0.1 * r(qr, scale=1.0)
        w["q_b_proj"] = r(H * (nope + rope), qr, scale=qr ** -0.5)
    else:
        w["q_proj"] = r(H * (nope + rope), hidden, scale=hidden ** -0.5)
        w["kv_a_proj_with_mqa"] = r(kr + rope, hidden, scale=hidden ** -0.5)
    return w


@pytest.mark.parametrize("dtn,H,qr,tol", [("bfloat16", 16, 1536, 4e-2), ("bfloat16", 8, None, 4e-2), ("float32", 16, None, 2e-4)])
def test_mla_module_prefill_then_absorbed_decode(dtn, H, qr, tol):
    from oracle import mla_ref
    from sglang_custom_amd.layers.attention.mi355_backend import Mi355AttnBackend
    from sglang_custom_amd.managers.schedule_batch import Req, ScheduleBatch
    from sglang_custom_amd.mem_cache.allocator import TokenToKVPoolAllocator
    from sglang_custom_amd.mem_cache.memory_pool import MLATokenToKVPool, ReqToTokenPool
    from sglang_custom_amd.model_executor.forward_batch_info import ForwardBatch
    from sglang_custom_amd.models.deepseek_v2 import DeepseekV2AttentionMLA

    dt = getattr(torch, dtn)
    hidden, nope, rope, vd, kr = 1024, 128, 64, 128, 512
    w = {k: v.to(dt) for k, v in _weights(hidden, H, nope, rope, vd, qr, kr, 3).items()}   # the oracle sees the rounded weights
    kvpool = MLATokenToKVPool(1024, 1, dt, kr, rope, 1, DEV)
    runner = SimpleNamespace(device=DEV, req_to_token_pool=ReqToTokenPool(8, 256, DEV), token_to_kv_pool=kvpool,
                             model_config=SimpleNamespace(num_attention_heads=H, num_key_value_heads=1, vocab_size=32),
                             tp_size=1, attn_backend=None)
    runner.attn_backend = Mi355AttnBackend(runner)
    alloc = TokenToKVPoolAllocator(1024, dt, DEV, kvpool)
    attn = DeepseekV2AttentionMLA(hidden, H, nope, rope, vd, qr, kr, 0, dt, DEV, max_position_embeddings=512)
    with torch.no_grad():
        for name, val in w.items():
            mod = getattr(attn, name)
            mod.weight.data.copy_(val.to(DEV))
    attn.post_load_weights()

    lens, steps = (37, 5, 64), 3
    g = torch.Generator().manual_seed(9)
    seqs = [torch.randn(n + steps, hidden, generator=g).to(dt) for n in lens]          # hidden states of every position
    want = [mla_ref.mla_attention(s, w, H, nope, rope, vd, qr, kr) for s in seqs]
    batch = ScheduleBatch.init_new([Req(i, [1] * n) for i, n in enumerate(lens)], runner.req_to_token_pool, alloc,
                                   model_config=runner.model_config)
    batch.prepare_for_extend()
    fb = ForwardBatch.init_new(batch.get_model_worker_batch(), runner)
    runner.attn_backend.init_forward_metadata(fb)
    x = torch.cat([s[:n] for s, n in zip(seqs, lens)]).to(DEV)
    with torch.no_grad():
        out = attn(fb.positions, x, fb).float().cpu()
    ref = torch.cat([r[:n] for r, n in zip(want, lens)]).float()
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() <= tol * scale, "prefill"
    # the cache holds [rmsnorm(c_kv) | rope(k_pe)] rows of width 576
    assert kvpool.get_key_buffer(0).shape[1:] == (1, kr + rope)
    for t in range(steps):
        batch.output_ids = torch.ones(len(lens), dtype=torch.int64, device=DEV)
        batch.prepare_for_decode()
        fb = ForwardBatch.init_new(batch.get_model_worker_batch(), runner)
        runner.attn_backend.init_forward_metadata(fb)
        x = torch.stack([s[n + t] for s, n in zip(seqs, lens)]).to(DEV)
        with torch.no_grad():
            out = attn(fb.positions, x, fb).float().cpu()
        ref = torch.stack([r[n + t] for r, n in zip(want, lens)]).float()
        assert fb.positions.tolist() == [n + t for n in lens]
        assert (out - ref).abs().max().item() <= tol * scale, f"decode step {t}"


@pytest.fixture(params=["per_token", "tiled"])
def mla_extend_mode(request):
    """Both forms of the absorbed extend over cached rows: one decode-style item per new token, and the
    kernel in which four tokens share a pass over the rows (forced on whatever the batch size)."""
    from sglang_custom_amd.layers.attention import mi355_backend as be

    keep = (be.MLA_EXTEND_TILED, be.MLA_EXTEND_TILED_MIN_WGS)
    be.MLA_EXTEND_TILED, be.MLA_EXTEND_TILED_MIN_WGS = request.param == "tiled", 0
    yield request.param
    be.MLA_EXTEND_TILED, be.MLA_EXTEND_TILED_MIN_WGS = keep


def test_mla_module_chunked_prefill_over_cached_latent_rows(mla_extend_mode):
    """A prompt prefilled in two chunks: the first through the multi-head path, the second as an
    extend over the cached latent rows in the absorbed form; then one decode step.  Every position
    equals the oracle's causal attention over the whole sequence."""
    from oracle import mla_ref
    from sglang_custom_amd.layers.attention.mi355_backend import Mi355AttnBackend
    from sglang_custom_amd.managers.schedule_batch import Req, ScheduleBatch
    from sglang_custom_amd.mem_cache.allocator import TokenToKVPoolAllocator
    from sglang_custom_amd.mem_cache.memory_pool import MLATokenToKVPool, ReqToTokenPool
    from sglang_custom_amd.model_executor.forward_batch_info import ForwardBatch
    from sglang_custom_amd.models.deepseek_v2 import DeepseekV2AttentionMLA

    dt, H, qr, tol = torch.bfloat16, 16, None, 4e-2
    hidden, nope, rope, vd, kr = 1024, 128, 64, 128, 512
    w = {k: v.to(dt) for k, v in _weights(hidden, H, nope, rope, vd, qr, kr, 5).items()}
    kvpool = MLATokenToKVPool(1024, 1, dt, kr, rope, 1, DEV)
    runner = SimpleNamespace(device=DEV, req_to_token_pool=ReqToTokenPool(8, 256, DEV), token_to_kv_pool=kvpool,
                             model_config=SimpleNamespace(num_attention_heads=H, num_key_value_heads=1, vocab_size=32),
                             tp_size=1, attn_backend=None)
    runner.attn_backend = Mi355AttnBackend(runner)
    alloc = TokenToKVPoolAllocator(1024, dt, DEV, kvpool)
    attn = DeepseekV2AttentionMLA(hidden, H, nope, rope, vd, qr, kr, 0, dt, DEV, max_position_embeddings=512)
    with torch.no_grad():
        for name, val in w.items():
            getattr(attn, name).weight.data.copy_(val.to(DEV))
    attn.post_load_weights()

    first, lens = (20, 3, 48), (45, 30, 49)            # tokens in chunk 1, prompt lengths
    g = torch.Generator().manual_seed(12)
    seqs = [torch.randn(n + 1, hidden, generator=g).to(dt) for n in lens]
    want = [mla_ref.mla_attention(s, w, H, nope, rope, vd, qr, kr).float() for s in seqs]
    scale = max(r.abs().max().item() for r in want)
    reqs = [Req(i, [1] * n) for i, n in enumerate(lens)]

    def run(batch, rows):
        fb = ForwardBatch.init_new(batch.get_model_worker_batch(), runner)
        runner.attn_backend.init_forward_metadata(fb)
        with torch.no_grad():
            return attn(fb.positions, rows.to(DEV), fb).float().cpu(), fb

    # chunk 1: no prefix -> multi-head path
    for r, f in zip(reqs, first):
        r.fill_ids = r.origin_input_ids[:f]
        r.extend_input_len = f
    batch = ScheduleBatch.init_new(reqs, runner.req_to_token_pool, alloc, model_config=runner.model_config)
    batch.prepare_for_extend()
    out, _ = run(batch, torch.cat([s[:f] for s, f in zip(seqs, first)]))
    assert (out - torch.cat([r[:f] for r, f in zip(want, first)])).abs().max().item() <= tol * scale
    # chunk 2: the rows of chunk 1 are the prefix -> absorbed extend over the cache
    table = runner.req_to_token_pool.req_to_token
    for r, f in zip(reqs, first):
        prefix = table[r.req_pool_idx, :f].clone()
        runner.req_to_token_pool.free(r.req_pool_idx)
        r.req_pool_idx = None
        r.init_next_round_input(prefix)
    batch = ScheduleBatch.init_new(reqs, runner.req_to_token_pool, alloc, model_config=runner.model_config)
    batch.prepare_for_extend()
    out, fb = run(batch, torch.cat([s[f:n] for s, f, n in zip(seqs, first, lens)]))
    assert fb.extend_prefix_lens_cpu == list(first)
    assert (out - torch.cat([r[f:n] for r, f, n in zip(want, first, lens)])).abs().max().item() <= tol * scale
    # one decode step on top
    batch.output_ids = torch.ones(len(lens), dtype=torch.int64, device=DEV)
    batch.prepare_for_decode()
    out, _ = run(batch, torch.stack([s[n] for s, n in zip(seqs, lens)]))
    assert (out - torch.stack([r[n] for r, n in zip(want, lens)])).abs().max().item() <= tol * scale


# ---- against vectors produced by the reference's own module and kernels (tests/golden/mla.npz) --------
@pytest.fixture(scope="module")
def golden_mla():
    import os

    import numpy as np

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mla.npz"))


@pytest.mark.parametrize("dtn", ["f32", "bf16", "f16"])
def test_mla_decode_kernel_matches_reference_vectors(golden_mla, dtn):
    """mi355_decode_attention_mla on the latent-cache decode cases the reference's torch-native
    backend was run on (16 / 22 / 128 heads, up to 1024 keys); within one output rounding."""
    import cases
    from sglang_custom_amd import ops

    tol = {"f32": 2e-5, "bf16": 1.6e-2, "f16": 2e-3}[dtn]
    R, P = cases.MLA_RANK, cases.MLA_ROPE
    for i, (B, Hq, _) in enumerate(cases.MLA_DECODE_CASES):
        q, kv, r2t, pool, lens = (x.to(DEV) for x in cases.mla_decode_inputs(i, cases.DTYPES[dtn]))
        want = torch.from_numpy(golden_mla[f"mla_decode/{dtn}/{i}"])
        for splits in (1, 4, 8):   # 8: some splits of the 33-key case hold 5 keys, less than a 16-key tile
            o = torch.empty((B, Hq, R), dtype=q.dtype, device=DEV)
            al = torch.empty((B, Hq, splits, R), dtype=torch.float32, device=DEV)
            ll = torch.empty((B, Hq, splits), dtype=torch.float32, device=DEV)
            ops.decode_attention_mla(q, kv, o, r2t, pool, lens, al, ll, splits, R, (R + P) ** -0.5)
            err = (o.float().cpu() - want).abs().max().item()
            assert err <= tol, (i, splits, err)


def _mla_world(dt, H, slots=1024):
    from sglang_custom_amd.layers.attention.mi355_backend import Mi355AttnBackend
    from sglang_custom_amd.mem_cache.allocator import TokenToKVPoolAllocator
    from sglang_custom_amd.mem_cache.memory_pool import MLATokenToKVPool, ReqToTokenPool

    kvpool = MLATokenToKVPool(slots, 1, dt, 512, 64, 1, DEV)
    runner = SimpleNamespace(device=DEV, req_to_token_pool=ReqToTokenPool(8, 256, DEV), token_to_kv_pool=kvpool,
                             model_config=SimpleNamespace(num_attention_heads=H, num_key_value_heads=1, vocab_size=32),
                             tp_size=1, attn_backend=None)
    runner.attn_backend = Mi355AttnBackend(runner)
    return runner, kvpool, TokenToKVPoolAllocator(slots, dt, DEV, kvpool)


@pytest.mark.parametrize("name,dtn,qr,scaling", _CASES.MLA_MODULE_CASES)
def test_mla_module_matches_reference_vectors(golden_mla, mla_extend_mode, name, dtn, qr, scaling):
    """The attention module on the GPU against the outputs of the reference's DeepseekV2AttentionMLA on
    the same weights and hidden states: prefill (multi-head form and absorbed form), the latent rows
    written to the cache, three decode steps, and a two-chunk prefill; with and without q_lora, with
    and without YaRN scaling (table magnitude and softmax scale)."""
    import cases
    from sglang_custom_amd.managers.schedule_batch import Req, ScheduleBatch
    from sglang_custom_amd.model_executor.forward_batch_info import ForwardBatch
    from sglang_custom_amd.models.deepseek_v2 import DeepseekV2AttentionMLA

    dt, d = cases.DTYPES[dtn], cases.MLA_DIMS
    g = lambda k: torch.from_numpy(golden_mla[f"mla_module/{name}/{k}"])   # noqa: E731
    tol = 2e-4 if dtn == "f32" else 4e-2          # of the output range: fp32 accumulation order / bf16 rounding chains
    H = d["heads"]
    attn = DeepseekV2AttentionMLA(d["hidden"], H, d["nope"], d["rope"], d["v"], qr, d["kv_lora"], 0, dt, DEV,
                                  rope_theta=cases.MLA_THETA, max_position_embeddings=4096, rms_norm_eps=cases.MLA_EPS,
                                  rope_scaling=scaling)
    with torch.no_grad():
        for wname, val in cases.mla_weights(qr).items():
            getattr(attn, wname).weight.data.copy_(val.to(dt).to(DEV))
    attn.post_load_weights()
    assert abs(attn.scaling - float(golden_mla[f"mla_module/{name}/softmax_scale"])) < 1e-12
    assert torch.equal(attn.rotary_emb.cos_sin_cache[:128].float().cpu(), g("cos_sin_cache_head"))
    scale = g("absorb_prefill").abs().max().item()

    def close(out, key):
        err = (out.float().cpu() - g(key)).abs().max().item()
        assert err <= tol * scale, (key, err, scale)

    def run(runner, batch, rows, method=None):
        fb = ForwardBatch.init_new(batch.get_model_worker_batch(), runner)
        runner.attn_backend.init_forward_metadata(fb)
        with torch.no_grad():
            return (method or attn)(fb.positions, rows.to(DEV), fb), fb

    lens, steps = cases.MLA_LENS, cases.MLA_STEPS
    seqs = cases.mla_hidden_states(lens, steps, 9, dt)
    x0 = torch.cat([s[:n] for s, n in zip(seqs, lens)])
    # prefill in the absorbed form (what the reference's backend ran), on its own pool
    runner, kvpool, alloc = _mla_world(dt, H)
    batch = ScheduleBatch.init_new([Req(i, [1] * n) for i, n in enumerate(lens)], runner.req_to_token_pool, alloc,
                                   model_config=runner.model_config)
    batch.prepare_for_extend()
    out, fb = run(runner, batch, x0, attn.forward_absorb)
    close(out, "absorb_prefill")
    # prefill through the module's own dispatch (multi-head form), then decode steps
    runner, kvpool, alloc = _mla_world(dt, H)
    batch = ScheduleBatch.init_new([Req(i, [1] * n) for i, n in enumerate(lens)], runner.req_to_token_pool, alloc,
                                   model_config=runner.model_config)
    batch.prepare_for_extend()
    out, fb = run(runner, batch, x0)
    close(out, "normal_prefill")
    rows_written = kvpool.get_key_buffer(0)[fb.out_cache_loc, 0].float().cpu()
    lat = g("normal_latent_cache")
    assert (rows_written - lat).abs().max().item() <= (1e-4 if dtn == "f32" else 3.2e-2) * lat.abs().max().item()
    for step in range(steps):
        batch.output_ids = torch.ones(len(lens), dtype=torch.int64, device=DEV)
        batch.prepare_for_decode()
        out, _ = run(runner, batch, torch.stack([s[n + step] for s, n in zip(seqs, lens)]))
        close(out, f"decode_{step}")
    # two chunks: the second extends over the cached latent rows
    first, clens = cases.MLA_CHUNK_FIRST, cases.MLA_CHUNK_LENS
    cseqs = cases.mla_hidden_states(clens, 1, 12, dt)
    runner, kvpool, alloc = _mla_world(dt, H)
    reqs = [Req(i, [1] * n) for i, n in enumerate(clens)]
    for r, f in zip(reqs, first):
        r.fill_ids = r.origin_input_ids[:f]
        r.extend_input_len = f
    batch = ScheduleBatch.init_new(reqs, runner.req_to_token_pool, alloc, model_config=runner.model_config)
    batch.prepare_for_extend()
    out, _ = run(runner, batch, torch.cat([s[:f] for s, f in zip(cseqs, first)]))
    close(out, "chunk1")
    table = runner.req_to_token_pool.req_to_token
    for r, f in zip(reqs, first):
        prefix = table[r.req_pool_idx, :f].clone()
        runner.req_to_token_pool.free(r.req_pool_idx)
        r.req_pool_idx = None
        r.init_next_round_input(prefix)
    batch = ScheduleBatch.init_new(reqs, runner.req_to_token_pool, alloc, model_config=runner.model_config)
    batch.prepare_for_extend()
    out, fb = run(runner, batch, torch.cat([s[f:n] for s, f, n in zip(cseqs, first, clens)]))
    assert fb.extend_prefix_lens_cpu == list(first)
    close(out, "chunk2")


def test_mla_module_with_fp8_kv_b_proj_matches_reference_vectors(golden_mla):
    """The absorbed (decode) path built from an fp8 kv_b_proj: tensor re-quantisation, w.to(bf16) * scale,
    then the same kernels -- against the reference module's outputs for that checkpoint."""
    import cases
    from sglang_custom_amd.managers.schedule_batch import Req, ScheduleBatch
    from sglang_custom_amd.model_executor.forward_batch_info import ForwardBatch
    from sglang_custom_amd.models.deepseek_v2 import DeepseekV2AttentionMLA

    name, dtn, qr, scaling = cases.MLA_FP8_CASE
    dt, d = cases.DTYPES[dtn], cases.MLA_DIMS
    g = lambda k: torch.from_numpy(golden_mla[f"mla_module/{name}/{k}"])   # noqa: E731
    H = d["heads"]
    attn = DeepseekV2AttentionMLA(d["hidden"], H, d["nope"], d["rope"], d["v"], qr, d["kv_lora"], 0, dt, DEV,
                                  rope_theta=cases.MLA_THETA, max_position_embeddings=4096, rms_norm_eps=cases.MLA_EPS)
    w = cases.mla_weights(qr)
    with torch.no_grad():
        for wname, val in w.items():
            getattr(attn, wname).weight.data.copy_(val.to(dt).to(DEV))
    wq, ws = cases.mla_kv_b_fp8(w["kv_b_proj"])
    attn.load_kv_b_proj_fp8(wq.to(DEV), ws.to(DEV))
    attn.post_load_weights()
    assert abs(float(attn.w_scale) - float(golden_mla[f"mla_module/{name}/w_scale"])) < 1e-9
    scale = g("absorb_prefill").abs().max().item()
    lens, steps = cases.MLA_LENS, cases.MLA_STEPS
    seqs = cases.mla_hidden_states(lens, steps, 9, dt)
    runner, kvpool, alloc = _mla_world(dt, H)
    batch = ScheduleBatch.init_new([Req(i, [1] * n) for i, n in enumerate(lens)], runner.req_to_token_pool, alloc,
                                   model_config=runner.model_config)

    def run(rows):
        fb = ForwardBatch.init_new(batch.get_model_worker_batch(), runner)
        runner.attn_backend.init_forward_metadata(fb)
        with torch.no_grad():
            return attn.forward_absorb(fb.positions, rows.to(DEV), fb).float().cpu()

    batch.prepare_for_extend()
    out = run(torch.cat([s[:n] for s, n in zip(seqs, lens)]))
    assert (out - g("absorb_prefill")).abs().max().item() <= 4e-2 * scale
    for step in range(steps):
        batch.output_ids = torch.ones(len(lens), dtype=torch.int64, device=DEV)
        batch.prepare_for_decode()
        out = run(torch.stack([s[n + step] for s, n in zip(seqs, lens)]))
        assert (out - g(f"decode_{step}")).abs().max().item() <= 4e-2 * scale, step


# ---- the tiled extend kernel on its own ---------------------------------------------------------------
@pytest.mark.parametrize("H", [16, 20, 128])
@pytest.mark.parametrize("dt,cap", [(torch.bfloat16, 0.0), (torch.float16, 0.0), (torch.bfloat16, 30.0)])
def test_mla_extend_tiled_kernel_vs_per_token_pass_and_fp64(H, dt, cap):
    """mi355_extend_attention_mla on a ragged extend batch (new-token counts that are not multiples of the
    4-token tile, a request without a prefix, a one-token request, prefixes that end inside a 32-key step)
    against one mi355_decode_attention_mla item per new token (the form the reference-vector tests pin) to one
    output rounding, and against float64 softmax math on every token of the shortest requests."""
    from sglang_custom_amd import ops

    R, P = 512, 64
    D = R + P
    new = [37, 1, 130, 4, 63]
    prefix = [0, 77, 300, 1, 33]
    B, width = len(new), 512
    g = torch.Generator(device=DEV).manual_seed(H + int(cap))
    slots = sum(n + p for n, p in zip(new, prefix)) + 1
    kv = torch.randn(slots, 1, D, device=DEV, generator=g).to(dt)
    perm = (1 + torch.randperm(slots - 1, device=DEV, generator=g)).to(torch.int32)
    r2t = torch.zeros(B + 3, width, dtype=torch.int32, device=DEV)
    rpi = torch.tensor([4, 0, 6, 2, 1], dtype=torch.int64, device=DEV)
    off = 0
    for b in range(B):
        n = new[b] + prefix[b]
        r2t[rpi[b], :n] = perm[off:off + n]
        off += n
    T = sum(new)
    q = (torch.randn(T, H, D, device=DEV, generator=g) * 0.5).to(dt)
    seq_lens = torch.tensor([n + p for n, p in zip(new, prefix)], dtype=torch.int64, device=DEV)
    qo = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    qo[1:] = torch.cumsum(torch.tensor(new, device=DEV), 0)
    o = torch.full((T, H, R), 7.0, dtype=dt, device=DEV)
    assert ops.extend_attention_mla_supported(H, R, P, dt)
    ops.extend_attention_mla(q, kv, o, r2t, rpi, seq_lens, qo, max(new), R, D ** -0.5, cap)
    # the per-token pass
    tok_req = torch.repeat_interleave(rpi, torch.tensor(new, device=DEV))
    tok_len = torch.cat([p + 1 + torch.arange(n) for n, p in zip(new, prefix)]).to(torch.int64).to(DEV)
    o2 = torch.empty_like(o)
    ops.decode_attention_mla(q, kv, o2, r2t, tok_req, tok_len, None, None, 1, R, D ** -0.5, cap)
    tol = 1.6e-2 if dt == torch.bfloat16 else 2e-3
    assert (o.float() - o2.float()).abs().max().item() <= tol
    # float64 on the requests with few keys
    for b in (0, 1, 3, 4):
        rows = kv[r2t[rpi[b], :new[b] + prefix[b]].long(), 0].double()
        for i in range(new[b]):
            t = int(qo[b]) + i
            s = (q[t].double() @ rows[:prefix[b] + i + 1].t()) * D ** -0.5
            if cap:
                s = cap * torch.tanh(s / cap)
            want = s.softmax(-1) @ rows[:prefix[b] + i + 1, :R]
            assert (o[t].double() - want).abs().max().item() <= tol, (b, i)


def test_mla_extend_tiled_refuses_other_latent_shapes():
    from sglang_custom_amd import ops
    from sglang_custom_amd._lib import Mi355Error

    assert not ops.extend_attention_mla_supported(16, 256, 64, torch.bfloat16)
    assert not ops.extend_attention_mla_supported(16, 512, 64, torch.float32)
    q = torch.zeros(4, 2, 320, device=DEV, dtype=torch.bfloat16)
    kv = torch.zeros(8, 1, 320, device=DEV, dtype=torch.bfloat16)
    o = torch.zeros(4, 2, 256, device=DEV, dtype=torch.bfloat16)
    r2t = torch.zeros(1, 8, dtype=torch.int32, device=DEV)
    with pytest.raises(Mi355Error):
        ops.extend_attention_mla(q, kv, o, r2t, torch.zeros(1, dtype=torch.int64, device=DEV),
                                 torch.full((1,), 4, dtype=torch.int64, device=DEV),
                                 torch.tensor([0, 4], dtype=torch.int32, device=DEV), 4, 256, 0.1)


@pytest.mark.parametrize("H", [16, 20, 128])
@pytest.mark.parametrize("dt,cap", [(torch.bfloat16, 0.0), (torch.float16, 0.0), (torch.bfloat16, 30.0)])
def test_mla_decode_over_fp8_latent_cache(H, dt, cap):
    """kv_cache_dtype fp8_e4m3 for the compressed cache (memory_pool.py:118-122, 769-909 of the reference):
    the rows are e4m3fn bytes, written by the scatter kernel (round to nearest even, saturating) and widened
    exactly when a fragment is used.  The fp8 LDS-ring kernel (16-byte aligned pointers) within one output rounding
    of float64 attention over the widened rows, its in-launch merge of the splits bit for bit its merge launch;
    the register kernel (selected by split scratch that is not 16-byte aligned) bit for bit the same kernel over
    a 16-bit cache that holds the widened values.  One and three KV splits."""
    from sglang_custom_amd import ops

    rank, rope = 512, 64
    D = rank + rope
    lens = [1, 15, 300, 65, 1029]
    total = sum(lens)
    rows = (_CASES.randn((total + 1, 1, D), 31, dt).float() * 3.0).to(dt).to(DEV)
    rows[5, 0, :4] = torch.tensor([1000.0, -1000.0, 448.0, 1e-3], dtype=dt, device=DEV)   # saturation and a tiny value
    cache8 = torch.zeros((total + 1, 1, D), dtype=torch.float8_e4m3fn, device=DEV)
    loc = torch.arange(total + 1, dtype=torch.int64, device=DEV)
    ops.set_kv_buffer(cache8, cache8, loc, rows, rows)
    wide = cache8.float()
    assert torch.isfinite(wide).all() and wide[5, 0, 0].item() == 448.0 and wide[5, 0, 1].item() == -448.0
    # round to nearest even at e4m3 precision: equal to torch's cast wherever the value is inside the range
    inside = rows.float().abs() <= 448.0
    assert torch.equal(wide[inside], rows.to(torch.float8_e4m3fn).float()[inside])
    cache16 = wide.to(dt)
    assert torch.equal(cache16.float(), wide)   # every e4m3 value is a 16-bit value
    r2t = torch.zeros((len(lens), max(lens)), dtype=torch.int32)
    perm = torch.randperm(total, generator=torch.Generator().manual_seed(4)) + 1
    pos = 0
    for b, n in enumerate(lens):
        r2t[b, :n] = perm[pos: pos + n].int()
        pos += n
    r2t = r2t.to(DEV)
    rpi = torch.arange(len(lens), dtype=torch.int64, device=DEV)
    lt = torch.tensor(lens, dtype=torch.int64, device=DEV)
    q = _CASES.randn((len(lens), H, D), 32, dt).to(DEV)
    scale = D ** -0.5
    for splits in (1, 3):
        n = len(lens) * H * max(splits, 2)
        lg = torch.empty(n * rank + 1, dtype=torch.float32, device=DEV)
        ls = torch.empty(n, dtype=torch.float32, device=DEV)
        got = torch.full((len(lens), H, rank), float("nan"), dtype=dt, device=DEV)
        ops.decode_attention_mla(q, cache8, got, r2t, rpi, lt, lg, ls, splits, rank, scale, cap)
        if splits > 1:
            tickets = torch.zeros(len(lens) * H, dtype=torch.int32, device=DEV)
            merged = torch.full_like(got, float("nan"))
            ops.decode_attention_mla(q, cache8, merged, r2t, rpi, lt, lg, ls, splits, rank, scale, cap, tickets=tickets)
            torch.cuda.synchronize()
            assert torch.equal(got.view(torch.int16), merged.view(torch.int16)) and not tickets.any(), (H, splits)
            reg8, same = torch.full_like(got, float("nan")), torch.full_like(got, float("nan"))
            ops.decode_attention_mla(q, cache8, reg8, r2t, rpi, lt, lg[1:], ls, splits, rank, scale, cap)
            ops.decode_attention_mla(q, cache16, same, r2t, rpi, lt, lg[1:], ls, splits, rank, scale, cap)
            torch.cuda.synchronize()
            assert torch.equal(reg8.view(torch.int16), same.view(torch.int16)), (H, splits)
        ring = torch.full_like(got, float("nan"))
        ops.decode_attention_mla(q, cache16, ring, r2t, rpi, lt, lg, ls, splits, rank, scale, cap)
        torch.cuda.synchronize()
        assert not torch.isnan(got.float()).any()
        # float64 attention over the widened rows
        for b, nkeys in enumerate(lens):
            kv = wide[r2t[b, :nkeys].long(), 0].double()
            sc = (q[b].double() @ kv.t()) * scale
            if cap > 0:
                sc = cap * torch.tanh(sc / cap)
            ref = torch.softmax(sc, -1) @ kv[:, :rank]
            tol = (2e-2 if dt == torch.bfloat16 else 3e-3) * max(ref.abs().max().item(), 1.0)
            assert (got[b].double() - ref).abs().max().item() <= tol, (H, splits, b)
            assert (ring[b].double() - ref).abs().max().item() <= tol


def test_fp8_latent_cache_refuses_other_shapes():
    from sglang_custom_amd import ops
    from sglang_custom_amd._lib import Mi355Error

    q = torch.zeros((1, 16, 128 + 64), dtype=torch.bfloat16, device=DEV)
    cache = torch.zeros((8, 1, 128 + 64), dtype=torch.float8_e4m3fn, device=DEV)
    o = torch.zeros((1, 16, 128), dtype=torch.bfloat16, device=DEV)
    r2t = torch.zeros((1, 4), dtype=torch.int32, device=DEV)
    with pytest.raises(Mi355Error):
        ops.decode_attention_mla(q, cache, o, r2t, torch.zeros(1, dtype=torch.int64, device=DEV),
                                 torch.ones(1, dtype=torch.int64, device=DEV), None, None, 1, 128, 0.1)


@pytest.mark.parametrize("H,splits,in_kernel_merge", [(16, 1, False), (16, 4, True), (20, 3, False), (128, 2, True), (5, 8, True)])
@pytest.mark.parametrize("dt,cs_dt,neox,cap", [(torch.bfloat16, torch.bfloat16, False, 0.0), (torch.float16, torch.float32, False, 0.0),
                                               (torch.bfloat16, torch.float32, True, 30.0)])
@pytest.mark.parametrize("kv8", [False, True], ids=["cache16", "cache_fp8"])
def test_mla_decode_with_rope_and_cache_write_inside_the_launch(H, splits, in_kernel_merge, dt, cs_dt, neox, cap, kv8):
    """mi355_decode_attention_mla_fused_rope (the reference's ROCm decode_attention_fwd_grouped_rope,
    rocm_mla_decode_rope.py:402-439, selected at deepseek_v2.py:973-980): q's rope part and k_pe unrotated in,
    one launch rotates them, writes the new latent row and attends.  Bit for bit the three-launch sequence
    rope_inplace -> set_kv_buffer -> decode_attention_mla: the outputs AND the cache rows; lengths that put the
    new token first / last in a tile and in either tile of a step, empty splits, head counts around the
    16-head tile, interleaved and neox pairing, fp32 and 16-bit tables; over a 16-bit cache and over an e4m3fn
    cache (the new row quantised inside the launch with the scatter kernel's arithmetic)."""
    from sglang_custom_amd import ops

    rank, rope = 512, 64
    D = rank + rope
    lens = [1, 16, 17, 32, 33, 300, 1029, 48]
    B, total = len(lens), sum(lens)
    cache0 = _CASES.randn((total + 1, 1, D), 41, dt).to(DEV)
    if kv8:
        rows16 = (cache0.float() * 3.0).to(dt)
        cache0 = torch.zeros((total + 1, 1, D), dtype=torch.float8_e4m3fn, device=DEV)
        ops.set_kv_buffer(cache0, cache0, torch.arange(total + 1, dtype=torch.int64, device=DEV), rows16, rows16)
    bits = torch.uint8 if kv8 else torch.int16
    r2t = torch.zeros((B, max(lens)), dtype=torch.int32)
    perm = torch.randperm(total, generator=torch.Generator().manual_seed(5)) + 1
    pos = 0
    for b, n in enumerate(lens):
        r2t[b, :n] = perm[pos: pos + n].int()
        pos += n
    r2t = r2t.to(DEV)
    rpi = torch.arange(B, dtype=torch.int64, device=DEV)
    lt = torch.tensor(lens, dtype=torch.int64, device=DEV)
    loc = torch.stack([r2t[b, n - 1] for b, n in enumerate(lens)]).to(torch.int64)
    positions = (lt - 1).clone()
    positions[3] = 4000          # the position need not be len - 1 (a cached prefix was dropped, or a test)
    q = _CASES.randn((B, H, D), 42, dt).to(DEV)
    lat = _CASES.randn((B, 192 + D), 43, dt).to(DEV)        # [q_a | kv_a | k_pe] as the fused projection leaves it
    k_nope = _CASES.randn((B, rank), 44, dt).to(DEV)          # the normed latent
    if kv8:
        k_nope[0, :3] = torch.tensor([1000.0, -1000.0, 1e-3], dtype=dt, device=DEV)   # saturation and a tiny value
    k_pe_view = lat[:, 192 + rank:]
    g = torch.Generator().manual_seed(6)
    cs = torch.cat([torch.randn(4096 + 8, rope // 2, generator=g).cos(), torch.randn(4096 + 8, rope // 2, generator=g).sin()],
                   dim=1).to(cs_dt).to(DEV).contiguous()
    scale = D ** -0.5
    n = B * H * max(splits, 1)
    lg = torch.empty(n * rank, dtype=torch.float32, device=DEV)
    ls = torch.empty(n, dtype=torch.float32, device=DEV)
    tickets = torch.zeros(B * H, dtype=torch.int32, device=DEV) if in_kernel_merge else None
    # three launches
    q_ref = q.clone()
    q_pe = q_ref[..., rank:].contiguous().view(B, -1)
    k_pe = k_pe_view.contiguous()
    ops.rope_inplace(positions, q_pe, k_pe, rope, cs, neox)
    q_ref[..., rank:] = q_pe.view(B, H, rope)
    cache_ref = cache0.clone()
    row = torch.cat([k_nope, k_pe], dim=-1).view(B, 1, D)
    ops.set_kv_buffer(cache_ref, cache_ref, loc, row, row)
    want = torch.full((B, H, rank), float("nan"), dtype=dt, device=DEV)
    ops.decode_attention_mla(q_ref, cache_ref, want, r2t, rpi, lt, lg, ls, splits, rank, scale, cap, tickets=tickets)
    torch.cuda.synchronize()
    # one launch
    for rep in range(2):
        cache = cache0.clone()
        got = torch.full_like(want, float("nan"))
        ops.decode_attention_mla_fused_rope(q, k_nope, k_pe_view, positions, loc, cs, neox, cache, got, r2t, rpi, lt, lg, ls,
                                            splits, rank, scale, cap, tickets=tickets)
        torch.cuda.synchronize()
        assert not torch.isnan(got.float()).any()
        assert torch.equal(cache.view(bits), cache_ref.view(bits)), "cache rows differ"
        bad = (got.view(torch.int16) != want.view(torch.int16)).any(-1).any(-1).nonzero().flatten().tolist()
        assert not bad, (bad, (got.float() - want.float()).abs().max().item())
        if tickets is not None:
            assert int(tickets.abs().sum()) == 0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,q_rank,kv_rank,rope", [(64, 1536, 512, 64), (1, 1536, 512, 64), (7, 0, 512, 64), (33, 2048, 1984, 64),
                                                      (5, 8, 8, 8)])
def test_mla_latent_norms_is_split_plus_two_rmsnorms_bit_for_bit(dt, rows, q_rank, kv_rank, rope):
    """One launch over the [q_a | c_kv | k_pe] row == split + contiguous + rmsnorm per part (and == the oracle's
    RMSNorm), from a dense row, from a row-strided view, and from fp32 split-K partials (bf16 and fp8-scaled)."""
    from oracle import ops_ref
    from sglang_custom_amd import ops

    g = torch.Generator().manual_seed(rows * 131 + q_rank + kv_rank)
    width = q_rank + kv_rank + rope
    wide = torch.randn(rows, width + 24, generator=g).to(dt).to(DEV)
    x = wide[:, 8: 8 + width]                       # row-strided, 16-byte aligned view
    qw = (1 + 0.1 * torch.randn(q_rank, generator=g)).to(dt).to(DEV) if q_rank else None
    kvw = (1 + 0.1 * torch.randn(kv_rank, generator=g)).to(dt).to(DEV)
    eps = 1e-6

    def unfused(row):
        q = ops.rmsnorm(row[:, :q_rank].contiguous(), qw, eps) if q_rank else None
        kv = ops.rmsnorm(row[:, q_rank: q_rank + kv_rank].contiguous(), kvw, eps)
        return q, kv, row[:, q_rank + kv_rank:].contiguous()

    def same(got, want):
        for a, b in zip(got, want):
            assert (a is None) == (b is None)
            if a is not None:
                assert a.is_contiguous() and torch.equal(a, b)

    want = unfused(x)
    same(ops.mla_latent_norms(qw, kvw, q_rank, kv_rank, rope, eps, x=x), want)
    same(ops.mla_latent_norms(qw, kvw, q_rank, kv_rank, rope, eps, x=x.contiguous()), want)
    ref_kv = ops_ref.rmsnorm(x[:, q_rank: q_rank + kv_rank].cpu(), kvw.cpu(), eps)
    torch.testing.assert_close(want[1].float().cpu(), ref_kv.float(), atol=2e-2, rtol=2e-2)

    if q_rank:   # + the per-token fp8 quantisation of the normed q_a rows (an fp8 q_b_proj's input)
        out5 = ops.mla_latent_norms(qw, kvw, q_rank, kv_rank, rope, eps, x=x, quant_q_fp8=True)
        same(out5[:3], want)
        q8, s8 = ops.per_token_quant_fp8(want[0])
        assert torch.equal(out5[3].view(torch.uint8), q8.view(torch.uint8)) and torch.equal(out5[4], s8)

    # split-K partials: summed in split order, rounded once, then normalised
    part = torch.randn(3, rows, width, generator=g).to(DEV)
    summed = ((part[0] + part[1]) + part[2]).to(dt)
    same(ops.mla_latent_norms(qw, kvw, q_rank, kv_rank, rope, eps, part=part), unfused(summed))
    xs = (0.5 + torch.rand(rows, 1, generator=g)).to(DEV)
    ws = (0.5 + torch.rand(width, 1, generator=g)).to(DEV)
    scaled = ((((part[0] + part[1]) + part[2]) * xs) * ws.view(1, -1)).to(dt)
    same(ops.mla_latent_norms(qw, kvw, q_rank, kv_rank, rope, eps, part=part, part_x_scale=xs, part_w_scale=ws), unfused(scaled))


def test_mla_latent_norms_refuses_what_it_does_not_serve():
    from sglang_custom_amd import _lib, ops

    kvw = torch.ones(512, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_lib.Mi355Error, match="mla_latent_norms"):
        ops.mla_latent_norms(None, kvw[:508].contiguous(), 0, 508, 64, 1e-6, x=torch.zeros(2, 572, dtype=torch.bfloat16, device=DEV))
    with pytest.raises(_lib.Mi355Error, match="mla_latent_norms"):
        ops.mla_latent_norms(torch.ones(4096, dtype=torch.bfloat16, device=DEV), kvw, 4096, 512, 64, 1e-6,
                             x=torch.zeros(2, 4096 + 576, dtype=torch.bfloat16, device=DEV))
    assert not ops.mla_latent_norms_supported(1536, 512, 64, torch.float32)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,heads,n,k,tail", [(64, 16, 512, 128, 64), (1, 16, 512, 128, 64), (17, 5, 512, 128, 64), (64, 16, 128, 512, 0),
                                              (33, 128, 128, 512, 0), (48, 3, 16, 32, 8)])
def test_mla_absorb_matches_float64_per_head_products(dt, m, heads, n, k, tail):
    """Both absorbed products of MLA decode (q_nope . w_kc with the rope part carried along; attn . w_vc in
    o_proj's layout) against a float64 product of the same rounded operands, rounded once; the copied tail is
    exact; row- and head-strided inputs."""
    from sglang_custom_amd import ops

    g = torch.Generator().manual_seed(m * 7 + heads * 3 + n + k)
    wide = torch.randn(m, heads, k + tail + 8, generator=g).to(dt).to(DEV)
    a = wide[..., : k + tail]
    w_t = (torch.randn(heads, n, k, generator=g) * k ** -0.5).to(dt).to(DEV)
    out = ops.mla_absorb(a, w_t, tail=tail)
    assert out.shape == (m, heads, n + tail) and out.is_contiguous()
    want = torch.einsum("mhk,hnk->mhn", a[..., :k].double().cpu(), w_t.double().cpu())
    # one rounding of an fp32 accumulation: within half a unit in the last place of the 16-bit type plus fp32 noise
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10    # spacing relative to the value, at most
    err = (out[..., :n].double().cpu() - want).abs()
    assert (err <= want.abs() * ulp * 0.51 + 1e-5).all(), err.max().item()
    if tail:
        assert torch.equal(out[..., n:], a[..., k:])
    torch.testing.assert_close(out[..., :n].float(), torch.bmm(a[..., :k].transpose(0, 1), w_t.transpose(1, 2)).transpose(0, 1).float(),
                               atol=4 * ulp, rtol=4 * ulp)


def test_mla_absorb_refuses_what_it_does_not_serve():
    from sglang_custom_amd import _lib, ops

    a = torch.zeros(65, 2, 128, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(2, 512, 128, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_lib.Mi355Error, match="mla_absorb"):
        ops.mla_absorb(a, w)
    with pytest.raises(_lib.Mi355Error, match="mla_absorb"):
        ops.mla_absorb(a[:4, :, :120], torch.zeros(2, 512, 120, dtype=torch.bfloat16, device=DEV))
    assert not ops.mla_absorb_supported(65, 512, 128, 64, torch.bfloat16)
    assert not ops.mla_absorb_supported(8, 512, 128, 64, torch.float32)


@pytest.mark.parametrize("bs,splits,fused_rope", [(128, 4, False), (256, 2, False), (128, 4, True), (64, 8, True)])
def test_mla_fp8_ring_two_splits_per_workgroup_on_ragged_batches(bs, splits, fused_rope):
    """Launches of at least 512 workgroups run the fp8 ring with TWO KV splits per workgroup (two teams of four waves
    whose results meet in LDS).  Ragged lengths: sequences shorter than one step (the second team of a workgroup has
    nothing to do), lengths around the 32-key step and 16-key tile edges, the new token in either team.  Every
    sequence against float64 attention over the widened rows; the in-launch merge bit for bit the merge launch; with
    RoPE + row write inside the launch bit for bit rope -> scatter -> decode, outputs and cache bytes."""
    from sglang_custom_amd import ops

    H, rank, rope = 16, 512, 64
    D = rank + rope
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(bs * 10 + splits)
    lens = torch.randint(1, 301, (bs,), generator=g).tolist()
    lens[:10] = [1, 2, 15, 16, 17, 31, 32, 33, 64, 65]
    total = sum(lens)
    rows16 = (torch.randn(total + 1, 1, D, generator=g) * 2.0).to(dt).to(DEV)
    cache0 = torch.zeros((total + 1, 1, D), dtype=torch.float8_e4m3fn, device=DEV)
    ops.set_kv_buffer(cache0, cache0, torch.arange(total + 1, dtype=torch.int64, device=DEV), rows16, rows16)
    r2t = torch.zeros((bs, max(lens)), dtype=torch.int32)
    perm = torch.randperm(total, generator=g) + 1
    pos = 0
    for b, n in enumerate(lens):
        r2t[b, :n] = perm[pos: pos + n].int()
        pos += n
    r2t = r2t.to(DEV)
    rpi = torch.arange(bs, dtype=torch.int64, device=DEV)
    lt = torch.tensor(lens, dtype=torch.int64, device=DEV)
    q = torch.randn(bs, H, D, generator=g).to(dt).to(DEV)
    scale = D ** -0.5
    lg = torch.empty(bs * H * splits * rank, dtype=torch.float32, device=DEV)
    ls = torch.empty(bs * H * splits, dtype=torch.float32, device=DEV)
    tickets = torch.zeros(bs * H, dtype=torch.int32, device=DEV)
    if not fused_rope:
        cache, q_used = cache0, q
        got = torch.full((bs, H, rank), float("nan"), dtype=dt, device=DEV)
        ops.decode_attention_mla(q, cache, got, r2t, rpi, lt, lg, ls, splits, rank, scale, tickets=tickets)
        again = torch.full_like(got, float("nan"))
        ops.decode_attention_mla(q, cache, again, r2t, rpi, lt, lg, ls, splits, rank, scale)      # merge launch
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), again.view(torch.int16)) and not tickets.any()
    else:
        loc = torch.stack([r2t[b, n - 1] for b, n in enumerate(lens)]).to(torch.int64)
        positions = (lt - 1).clone()
        k_nope = torch.randn(bs, rank, generator=g).to(dt).to(DEV)
        k_pe_raw = torch.randn(bs, rope, generator=g).to(dt).to(DEV)
        cs = torch.cat([torch.randn(512, rope // 2, generator=g).cos(), torch.randn(512, rope // 2, generator=g).sin()], dim=1).to(DEV).contiguous()
        q_used = q.clone()
        q_pe = q_used[..., rank:].contiguous().view(bs, -1)
        k_pe = k_pe_raw.clone()
        ops.rope_inplace(positions, q_pe, k_pe, rope, cs, False)
        q_used[..., rank:] = q_pe.view(bs, H, rope)
        cache = cache0.clone()
        row = torch.cat([k_nope, k_pe], dim=-1).view(bs, 1, D)
        ops.set_kv_buffer(cache, cache, loc, row, row)
        want = torch.full((bs, H, rank), float("nan"), dtype=dt, device=DEV)
        ops.decode_attention_mla(q_used, cache, want, r2t, rpi, lt, lg, ls, splits, rank, scale, tickets=tickets)
        cache_f = cache0.clone()
        got = torch.full_like(want, float("nan"))
        ops.decode_attention_mla_fused_rope(q, k_nope, k_pe_raw, positions, loc, cs, False, cache_f, got, r2t, rpi, lt, lg, ls,
                                            splits, rank, scale, tickets=tickets)
        torch.cuda.synchronize()
        assert torch.equal(cache_f.view(torch.uint8), cache.view(torch.uint8)), "cache rows differ"
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)) and not tickets.any()
    assert not torch.isnan(got.float()).any()
    wide = cache.float()
    worst = 0.0
    for b, nkeys in enumerate(lens):
        kv = wide[r2t[b, :nkeys].long(), 0].double()
        ref = torch.softmax((q_used[b].double() @ kv.t()) * scale, -1) @ kv[:, :rank]
        err = (got[b].double() - ref).abs().max().item() / max(ref.abs().max().item(), 1.0)
        worst = max(worst, err)
        assert err <= 2e-2, (b, nkeys, err)


@pytest.mark.parametrize("T,H,nope,rope,vd,rank", [(37, 16, 128, 64, 128, 512), (1, 128, 128, 64, 128, 512), (300, 3, 64, 32, 40, 128),
                                                   (5, 2, 8, 8, 8, 8), (0, 16, 128, 64, 128, 512)])
@pytest.mark.parametrize("dt,cs_dt,neox", [(torch.bfloat16, torch.bfloat16, False), (torch.float16, torch.float32, False),
                                           (torch.bfloat16, torch.float32, True)])
@pytest.mark.parametrize("kv8", [False, True], ids=["cache16", "cache_fp8"])
def test_mla_prefill_assemble_is_rope_copies_and_cache_write_bit_for_bit(T, H, nope, rope, vd, rank, dt, cs_dt, neox, kv8):
    """mi355_mla_prefill_assemble == the glue of MLA's multi-head prefill (deepseek_v2.py:1131-1179) launch by
    launch: rope_inplace on (q's rope part, k_pe), k = [kv nope | k_pe per head], v = kv's value part,
    set_kv_buffer of [c_kv | k_pe].  q, k, v and the cache rows, bit for bit; k_pe and c_kv as row-strided views
    of the fused projection's output, scattered slots, 16-bit and e4m3fn caches, saturating values."""
    from sglang_custom_amd import ops

    assert ops.mla_prefill_assemble_supported(nope, rope, vd, rank, dt)
    slots = 2 * T + 3
    q = _CASES.randn((T, H, nope + rope), 51, dt).to(DEV)
    kv = _CASES.randn((T, H, nope + vd), 52, dt).to(DEV)
    lat = _CASES.randn((T, 24 + rank + rope), 53, dt).to(DEV)
    k_pe_view = lat[:, 24 + rank:]
    c_kv_all = _CASES.randn((T, rank + 8), 54, dt).to(DEV)
    c_kv = c_kv_all[:, :rank]
    if kv8 and T:
        c_kv[0, :3] = torch.tensor([1000.0, -1000.0, 1e-3], dtype=dt, device=DEV)
    g = torch.Generator().manual_seed(7)
    cs = torch.cat([torch.randn(5000, rope // 2, generator=g).cos(), torch.randn(5000, rope // 2, generator=g).sin()],
                   dim=1).to(cs_dt).to(DEV).contiguous()
    positions = torch.randint(0, 5000, (T,), generator=g, dtype=torch.int64).to(DEV)
    loc = (torch.randperm(slots - 1, generator=g)[:T] + 1).to(torch.int64).to(DEV)
    cdt = torch.float8_e4m3fn if kv8 else dt
    bits = torch.uint8 if kv8 else torch.int16
    cache0 = torch.zeros((slots, 1, rank + rope), dtype=cdt, device=DEV)
    # launch by launch
    q_ref = q.clone()
    k_pe = k_pe_view.clone()             # (contiguous() of a one-row view is the view itself)
    if T:
        q_pe = q_ref[..., nope:].contiguous().view(T, -1)
        ops.rope_inplace(positions, q_pe, k_pe, rope, cs, neox)
        q_ref[..., nope:] = q_pe.view(T, H, rope)
    k_ref = torch.empty_like(q_ref)
    k_ref[..., :nope] = kv[..., :nope]
    k_ref[..., nope:] = k_pe.view(T, 1, rope)
    v_ref = kv[..., nope:].contiguous()
    cache_ref = cache0.clone()
    if T:
        row = torch.cat([c_kv, k_pe], dim=-1).view(T, 1, rank + rope)
        ops.set_kv_buffer(cache_ref, cache_ref, loc, row, row)
    # one launch
    cache = cache0.clone()
    q_got = q.clone()
    k_got, v_got = ops.mla_prefill_assemble(q_got, kv, k_pe_view, c_kv, positions, cs, neox, cache, loc, nope, rope, vd)
    torch.cuda.synchronize()
    assert torch.equal(q_got.view(torch.int16), q_ref.view(torch.int16))
    bad = (k_got.view(torch.int16) != k_ref.view(torch.int16)).nonzero()
    assert bad.numel() == 0, ("k differs at (token, head, feature)", bad[:8].tolist(), len(bad))
    assert torch.equal(v_got.view(torch.int16), v_ref.view(torch.int16))
    assert torch.equal(cache.view(bits), cache_ref.view(bits))


def test_mla_prefill_assemble_refuses_what_it_does_not_serve():
    from sglang_custom_amd import ops
    from sglang_custom_amd._lib import Mi355Error

    assert not ops.mla_prefill_assemble_supported(128, 64, 128, 512, torch.float32)
    assert not ops.mla_prefill_assemble_supported(128, 60, 128, 512, torch.bfloat16)
    assert not ops.mla_prefill_assemble_supported(128, 512, 128, 512, torch.bfloat16)
    dt = torch.bfloat16
    q = torch.zeros((2, 2, 128 + 36), dtype=dt, device=DEV)
    kv = torch.zeros((2, 2, 256), dtype=dt, device=DEV)
    cache = torch.zeros((4, 1, 512 + 36), dtype=dt, device=DEV)
    i64 = torch.zeros(2, dtype=torch.int64, device=DEV)
    cs = torch.zeros((8, 36), dtype=torch.float32, device=DEV)
    with pytest.raises(Mi355Error):
        ops.mla_prefill_assemble(q, kv, torch.zeros((2, 36), dtype=dt, device=DEV), torch.zeros((2, 512), dtype=dt, device=DEV),
                                 i64, cs, False, cache, i64, 128, 36, 128)
