"""DeepSeek multi-head latent attention as one module on the GPU: prefill through the multi-head
path (latent expanded by kv_b_proj, latent rows written to the compressed cache), then decode steps
through the absorbed path (multi-query attention over the 576-wide cache rows on the matrix cores),
checked at every position against the oracle's float64 causal attention without absorption."""
from types import SimpleNamespace

import pytest
import torch

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
