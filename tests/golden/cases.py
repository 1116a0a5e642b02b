"""Deterministic inputs for the golden vectors.  Imported by gen_golden.py (which feeds them to
the reference) and by the tests (which feed them to the oracle and to the HIP kernels), so the
fixtures only need to store OUTPUTS.  torch's CPU generator is bit-reproducible for a given
torch build; the build container and the GPU box run the same image."""
import numpy as np
import torch

DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def _g(seed):
    return torch.Generator().manual_seed(seed)


def randn(shape, seed, dtype, scale=1.0):
    return (scale * torch.randn(shape, generator=_g(seed), dtype=torch.float32)).to(dtype)


NORM_CASES = [(4, 512), (3, 111), (2, 4096), (5, 500)]          # (rows, hidden)
ACT_CASES = [(4, 128), (3, 1000), (2, 3584)]                   # (rows, d)
# (tokens, hq, hkv, head, rotary_dim, neox, llama3-scaling)
ROPE_CASES = [(9, 8, 2, 64, 64, True, False), (5, 4, 4, 128, 64, False, False),
              (7, 32, 8, 128, 128, True, True)]
LLAMA3 = dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
              original_max_position_embeddings=256)
# (B, Hq, Hkv, D, max_len)
DECODE_CASES = [(3, 8, 2, 64, 50), (2, 4, 4, 128, 70), (2, 32, 8, 128, 33), (1, 16, 1, 128, 129)]
# (Hq, Hkv, D, prefix_lens, extend_lens)
EXTEND_CASES = [(8, 2, 64, [0, 0, 0], [5, 17, 32]), (8, 2, 128, [40, 0, 13], [30, 70, 3]),
                (4, 4, 128, [20], [1]), (16, 1, 128, [7, 0], [20, 65])]
SAMPLING_CASES = [(4, 111), (3, 2048), (2, 8192)]              # (B, vocab)


def norm_inputs(i, dt):
    rows, h = NORM_CASES[i]
    return (randn((rows, h), 100 + i, dt), randn((rows, h), 200 + i, dt),
            (1 + 0.1 * torch.randn(h, generator=_g(300 + i), dtype=torch.float32)).to(dt))


def act_inputs(i, dt):
    rows, d = ACT_CASES[i]
    return randn((rows, 2 * d), 400 + i, dt)


def rope_inputs(i, dt):
    t, hq, hkv, hd, rd, neox, l3 = ROPE_CASES[i]
    pos = torch.randint(0, 2048, (t,), generator=_g(500 + i))
    return pos, randn((t, hq * hd), 510 + i, dt), randn((t, hkv * hd), 520 + i, dt)


def paged_inputs(seed, dt, B, Hkv, D, lens, extra_slots=7):
    """A scrambled paged KV cache: returns k_cache, v_cache, req_to_token, req_pool_indices."""
    total = int(sum(lens))
    slots = total + extra_slots
    g = _g(seed)
    k_cache = randn((slots + 1, Hkv, D), seed + 1, dt)
    v_cache = randn((slots + 1, Hkv, D), seed + 2, dt)
    perm = torch.randperm(slots, generator=g) + 1
    max_len = int(max(lens))
    req_pool_indices = torch.randperm(B + 3, generator=g)[:B].to(torch.int64)
    req_to_token = torch.zeros((B + 3, max_len + 4), dtype=torch.int32)
    p = 0
    for b in range(B):
        req_to_token[req_pool_indices[b], : lens[b]] = perm[p: p + lens[b]].to(torch.int32)
        p += lens[b]
    return k_cache, v_cache, req_to_token, req_pool_indices


def decode_inputs(i, dt):
    B, Hq, Hkv, D, max_len = DECODE_CASES[i]
    g = _g(600 + i)
    lens = torch.randint(max(1, max_len // 2), max_len + 1, (B,), generator=g)
    lens[0] = max_len
    k_cache, v_cache, r2t, pool = paged_inputs(610 + 10 * i, dt, B, Hkv, D, lens.tolist())
    q = randn((B, Hq, D), 605 + i, dt)
    return q, k_cache, v_cache, r2t, pool, lens.to(torch.int64)


def extend_inputs(i, dt):
    Hq, Hkv, D, pre, ext = EXTEND_CASES[i]
    B = len(pre)
    seq = [p + e for p, e in zip(pre, ext)]
    k_cache, v_cache, r2t, pool = paged_inputs(710 + 10 * i, dt, B, Hkv, D, seq)
    T = sum(ext)
    q = randn((T, Hq, D), 705 + i, dt)
    return (q, k_cache, v_cache, r2t, pool, torch.tensor(seq, dtype=torch.int64),
            torch.tensor(pre, dtype=torch.int64), torch.tensor(ext, dtype=torch.int64))


def sampling_inputs(i):
    B, V = SAMPLING_CASES[i]
    g = _g(800 + i)
    logits = 3.0 * torch.randn((B, V), generator=g, dtype=torch.float32)
    temps = torch.rand((B, 1), generator=g, dtype=torch.float32) + 0.5
    top_ks = torch.randint(1, 40, (B,), generator=g).to(torch.int32)
    top_ps = torch.rand((B,), generator=g, dtype=torch.float32) * 0.8 + 0.1
    min_ps = torch.rand((B,), generator=g, dtype=torch.float32) * 0.05
    return logits, temps, top_ks, top_ps, min_ps


# end-to-end: (name, preset, dtype, prompt lengths, decode steps)
E2E_CASES = [
    ("tiny64_f32", "tiny-llama-d64", "f32", (5, 17, 32), 8),
    ("tiny64_bf16", "tiny-llama-d64", "bf16", (5, 17, 32), 8),
    ("tiny128_f32", "tiny-llama-d128", "f32", (3, 40), 8),
    ("tiny128_bf16", "tiny-llama-d128", "bf16", (3, 40), 8),
]
# configuration 1 at its real shape: Llama-3.2-1B (16 layers, vocab 128256, tied head, llama3 rope
# scaling), one request, 128-token prompt, 128 greedy tokens.  The fixture keeps the tokens, the
# prefill and last-step logits and the five largest logits of every step (not all 128 x 128256).
E2E_LONG_CASES = [("llama1b_bf16", "llama-3.2-1b", "bf16", (128,), 128), ("llama1b_f32", "llama-3.2-1b", "f32", (128,), 128)]
E2E_PREFIX_CUT = 4   # second scenario: first cache prompt[:4], then extend with the rest


def e2e_prompts(preset_vocab, lens, seed=0):
    rng = np.random.RandomState(seed)
    return [rng.randint(0, preset_vocab, n).tolist() for n in lens]


# ---- DeepSeek multi-head latent attention (compressed KV: 512-wide latent + 64 rope features) ----
# decode over a latent cache [slots, 1, 576] whose value is the row's first 512 features:
# (B, Hq, max_len); the MLA shapes of the reference's kernel tests (test/srt/cpu/test_decode.py:62-163,
# test/srt/test_triton_attention_kernels.py:293-388) plus the TP=8 head count of DeepSeek-V3
MLA_DECODE_CASES = [(2, 16, 100), (3, 128, 257), (1, 22, 1024), (4, 16, 33)]
MLA_RANK, MLA_ROPE = 512, 64


def mla_decode_inputs(i, dt):
    B, Hq, max_len = MLA_DECODE_CASES[i]
    g = _g(1300 + i)
    lens = torch.randint(max(1, max_len // 2), max_len + 1, (B,), generator=g)
    lens[0] = max_len
    kv_cache, _, r2t, pool = paged_inputs(1310 + 10 * i, dt, B, 1, MLA_RANK + MLA_ROPE, lens.tolist())
    q = randn((B, Hq, MLA_RANK + MLA_ROPE), 1305 + i, dt)
    return q, kv_cache, r2t, pool, lens.to(torch.int64)


# the attention MODULE: (name, dtype, q_lora_rank, rope_scaling)
YARN_V3 = dict(type="yarn", factor=40.0, original_max_position_embeddings=4096, beta_fast=32, beta_slow=1,
               mscale=1.0, mscale_all_dim=1.0)              # DeepSeek-V3's published rope_scaling
YARN_ALT = dict(type="yarn", factor=8.0, original_max_position_embeddings=512, beta_fast=32, beta_slow=1,
                mscale=1.0, mscale_all_dim=0.5)             # cos/sin magnitude scale != 1
MLA_MODULE_CASES = [("f32_plain", "f32", None, None), ("bf16_qlora", "bf16", 192, None),
                    ("f32_qlora_yarn", "f32", 192, YARN_V3), ("bf16_yarn_alt", "bf16", None, YARN_ALT)]
# a fifth scenario, "bf16_fp8_kv_b": kv_b_proj stored as e4m3fn with per-output-channel scales (an fp8
# W8A8 checkpoint); the absorbed path then runs on w_kc / w_vc re-quantised per tensor and widened to
# bf16 times the tensor scale (models/deepseek_v2.py:2196-2252, 2271-2297, 1242-1247, 1300-1306)
MLA_FP8_CASE = ("bf16_fp8_kv_b", "bf16", None, None)
MLA_DIMS = dict(hidden=512, heads=16, nope=128, rope=64, v=128, kv_lora=512)
MLA_THETA, MLA_EPS = 10000.0, 1e-6
MLA_LENS, MLA_STEPS = (37, 5, 64), 3          # prompt lengths, decode steps
MLA_CHUNK_FIRST, MLA_CHUNK_LENS = (20, 3, 48), (45, 30, 49)   # tokens in chunk 1, prompt lengths


def mla_weights(q_lora_rank, seed=3):
    """fp32 master weights of one attention module (rounded to the case dtype by the caller)."""
    d = MLA_DIMS
    hidden, H, nope, rope, vd, kr = d["hidden"], d["heads"], d["nope"], d["rope"], d["v"], d["kv_lora"]
    g = _g(seed)

    def r(*shape, scale):
        return torch.randn(*shape, generator=g) * scale

    w = {"kv_a_layernorm": 1 + 0.1 * r(kr, scale=1.0), "kv_b_proj": r(H * (nope + vd), kr, scale=kr ** -0.5),
         "o_proj": r(hidden, H * vd, scale=(H * vd) ** -0.5)}
    if q_lora_rank is not None:
        w["fused_qkv_a_proj_with_mqa"] = r(q_lora_rank + kr + rope, hidden, scale=hidden ** -0.5)
        w["q_a_layernorm"] = 1 + 0.1 * r(q_lora_rank, scale=1.0)
        w["q_b_proj"] = r(H * (nope + rope), q_lora_rank, scale=q_lora_rank ** -0.5)
    else:
        w["q_proj"] = r(H * (nope + rope), hidden, scale=hidden ** -0.5)
        w["kv_a_proj_with_mqa"] = r(kr + rope, hidden, scale=hidden ** -0.5)
    return w


def mla_kv_b_fp8(w_kv_b):
    """Per-output-channel e4m3fn quantisation of kv_b_proj as an fp8 checkpoint stores it:
    returns (weight fp8 [N, K], weight_scale fp32 [N, 1])."""
    w = w_kv_b.float()
    scale = w.abs().amax(1, keepdim=True).clamp(min=1e-12) / 448.0
    return (w / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn), scale


def mla_hidden_states(lens, extra, seed, dt):
    """Hidden states of every position of every sequence (prompt + `extra` decode positions)."""
    g = _g(seed)
    return [torch.randn(n + extra, MLA_DIMS["hidden"], generator=g).to(dt) for n in lens]


def mla_slots(total, seed):
    """A scrambled assignment of KV slots 1..total (slot 0 is the padding sink)."""
    return torch.randperm(total, generator=_g(seed)) + 1


# ---- input-token log-probabilities of a prefill (tests/golden/logprob.json, gen_logprob_golden.py) ----
def run_logprob_scenarios(mr, golden):
    """Runs the two scenarios of logprob.json (plain prefill; extend over a cached prefix) on ModelRunner
    `mr` of this package and returns records shaped like the golden's."""
    import torch

    from sglang_custom_amd import bench_one_batch as bob
    from sglang_custom_amd.managers.schedule_batch import Req, ScheduleBatch
    from sglang_custom_amd.model_executor.forward_batch_info import ForwardBatch

    prompts = e2e_prompts(mr.model_config.vocab_size, golden["lens"])

    def run(reqs):
        batch = ScheduleBatch.init_new(reqs=reqs, req_to_token_pool=mr.req_to_token_pool,
                                       token_to_kv_pool_allocator=mr.token_to_kv_pool_allocator, tree_cache=None,
                                       model_config=mr.model_config)
        batch.prepare_for_extend()
        fb = ForwardBatch.init_new(batch.get_model_worker_batch(), mr)
        with torch.no_grad():
            lo, _ = mr.forward(fb)
        return lo, batch

    def record(lo, batch):
        return dict(extend_logprob_start_lens=list(batch.extend_logprob_start_lens),
                    extend_input_logprob_token_ids=batch.extend_input_logprob_token_ids.tolist(),
                    next_token_logits_top1=lo.next_token_logits.argmax(-1).tolist(),
                    next_token_logits_max=lo.next_token_logits.max(-1).values.tolist(),
                    input_token_logprobs=lo.input_token_logprobs.tolist(),
                    input_top_logprobs_val=lo.input_top_logprobs_val, input_top_logprobs_idx=lo.input_top_logprobs_idx)

    out = {}
    g = golden["prefill"]
    bob.clear_pools(mr)
    lo, batch = run([Req(i, p, return_logprob=True, top_logprobs_num=g["top_logprobs_nums"][i],
                         logprob_start_len=g["logprob_start_lens"][i]) for i, p in enumerate(prompts)])
    out["prefill"] = record(lo, batch)
    g = golden["prefix_extend"]
    cut = g["cut"]
    bob.clear_pools(mr)
    reqs = [Req(i, p[:cut]) for i, p in enumerate(prompts)]
    run(reqs)
    for i, r in enumerate(reqs):
        r.fill_ids = r.fill_ids + prompts[i][cut:]
        r.origin_input_ids = list(r.fill_ids)
        r.prefix_indices = mr.req_to_token_pool.req_to_token[i, :cut].clone()
        r.extend_input_len = len(r.fill_ids) - cut
        r.return_logprob, r.logprob_start_len, r.top_logprobs_num = True, g["logprob_start_lens"][i], g["top_logprobs_nums"][i]
    mr.req_to_token_pool.clear()
    lo, batch = run(reqs)
    out["prefix_extend"] = record(lo, batch)
    return out


def check_logprob_records(got, golden, tol):
    for name in ("prefill", "prefix_extend"):
        a, b = got[name], golden[name]
        assert a["extend_logprob_start_lens"] == b["extend_logprob_start_lens"], name
        assert a["extend_input_logprob_token_ids"] == b["extend_input_logprob_token_ids"], name
        assert a["next_token_logits_top1"] == b["next_token_logits_top1"], name
        assert max(abs(x - y) for x, y in zip(a["next_token_logits_max"], b["next_token_logits_max"])) < tol, name
        assert len(a["input_token_logprobs"]) == len(b["input_token_logprobs"]), name
        assert max(abs(x - y) for x, y in zip(a["input_token_logprobs"], b["input_token_logprobs"])) < tol, name
        assert len(a["input_top_logprobs_val"]) == len(b["input_top_logprobs_val"]), name
        for ra, rb, ia, ib in zip(a["input_top_logprobs_val"], b["input_top_logprobs_val"],
                                  a["input_top_logprobs_idx"], b["input_top_logprobs_idx"]):
            assert len(ra) == len(rb), name
            for va, vb, xa, xb in zip(ra, rb, ia, ib):
                assert len(va) == len(vb) and max((abs(x - y) for x, y in zip(va, vb)), default=0.0) < tol, name
                # ids agree wherever the reference's neighbouring values are further apart than the tolerance
                for j, (x, y) in enumerate(zip(xa, xb)):
                    tie = any(abs(vb[j] - vb[k]) < 2 * tol for k in range(len(vb)) if k != j)
                    assert x == y or tie, (name, xa, xb)
