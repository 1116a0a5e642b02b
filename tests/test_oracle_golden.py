"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference's own code
(tests/golden/gen_golden.py).  Same torch build, same primitive ops => bit-exact for the ops;
the end-to-end runs must give identical greedy tokens and near-identical logits."""
import numpy as np
import pytest
import torch

import cases
from conftest import load_e2e
from oracle import ops_ref
from oracle.llama_ref import OracleLlama
from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights

DT = cases.DTYPES


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("dtn", list(DT))
def test_rmsnorm_matches_reference(golden_ops, dtn):
    dt = DT[dtn]
    for i in range(len(cases.NORM_CASES)):
        x, r, w = cases.norm_inputs(i, dt)
        assert torch.equal(ops_ref.rmsnorm(x, w, 1e-5).float(), t(golden_ops[f"rmsnorm/{dtn}/{i}"]))
        y, r2 = ops_ref.rmsnorm(x, w, 1e-5, r)
        assert torch.equal(y.float(), t(golden_ops[f"fused_add_rmsnorm/{dtn}/{i}/x"]))
        assert torch.equal(r2.float(), t(golden_ops[f"fused_add_rmsnorm/{dtn}/{i}/residual"]))


@pytest.mark.parametrize("dtn", list(DT))
def test_silu_and_mul_matches_reference(golden_ops, dtn):
    for i in range(len(cases.ACT_CASES)):
        got = ops_ref.silu_and_mul(cases.act_inputs(i, DT[dtn]))
        assert torch.equal(got.float(), t(golden_ops[f"silu_and_mul/{dtn}/{i}"]))


@pytest.mark.parametrize("dtn", list(DT))
def test_rope_matches_reference(golden_ops, dtn):
    dt = DT[dtn]
    for i, (T, hq, hkv, hd, rd, neox, l3) in enumerate(cases.ROPE_CASES):
        pos, q, k = cases.rope_inputs(i, dt)
        cache = ops_ref.cos_sin_cache(rd, 2048, 500000.0 if l3 else 10000.0, cases.LLAMA3 if l3 else None, dt)
        if dtn == "f32":
            assert torch.equal(cache[::37], t(golden_ops[f"rope_cache/{i}"]))
        q2, k2 = ops_ref.apply_rope(pos, q, k, hd, cache, neox)
        assert torch.equal(q2.float(), t(golden_ops[f"rope/{dtn}/{i}/q"]))
        assert torch.equal(k2.float(), t(golden_ops[f"rope/{dtn}/{i}/k"]))


@pytest.mark.parametrize("dtn", list(DT))
def test_decode_attention_matches_reference(golden_ops, dtn):
    for i, (B, Hq, Hkv, D, _) in enumerate(cases.DECODE_CASES):
        q, kc, vc, r2t, pool, lens = cases.decode_inputs(i, DT[dtn])
        o = ops_ref.decode_attention(q, kc, vc, r2t, pool, lens, D ** -0.5, Hq != Hkv)
        assert torch.equal(o.float(), t(golden_ops[f"decode/{dtn}/{i}"]))
        # independent check of the SDPA-based oracle with an fp64 softmax
        for b in range(B):
            toks = r2t[pool[b], : lens[b]].long()
            ref = ops_ref.attention_fp64(q[b][:, None], kc[toks].transpose(0, 1), vc[toks].transpose(0, 1), D ** -0.5)[:, 0]
            tol = 1e-5 if dtn == "f32" else 2e-2
            assert (o[b].double() - ref).abs().max() < tol
        if dtn in ("bf16", "f16"):
            # the reference's C++ AMX kernel vs its torch-native path, held to the cosine bar of
            # the reference's own test (test/srt/cpu/test_decode.py:139-143); its absolute error on
            # these GQA cases is up to 0.28, so the torch-native path is the oracle, not the kernel
            ck = t(golden_ops[f"decode_cpu_kernel/{dtn}/{i}"])
            cos = torch.nn.functional.cosine_similarity(ck.flatten(), o.float().flatten(), dim=0)
            assert cos > 0.99


@pytest.mark.parametrize("dtn", list(DT))
def test_extend_attention_matches_reference(golden_ops, dtn):
    for i, (Hq, Hkv, D, pre, ext) in enumerate(cases.EXTEND_CASES):
        q, kc, vc, r2t, pool, seq, pre_t, ext_t = cases.extend_inputs(i, DT[dtn])
        o = ops_ref.extend_attention(q, kc, vc, r2t, pool, seq, pre_t, ext_t, D ** -0.5, Hq != Hkv, True)
        assert torch.equal(o.float(), t(golden_ops[f"extend/{dtn}/{i}"]))


def test_sampling_matches_reference(golden_ops):
    for i in range(len(cases.SAMPLING_CASES)):
        logits, temps, ks, ps, mps = cases.sampling_inputs(i)
        probs = ops_ref.softmax_temperature(logits, temps)
        assert torch.equal(probs, t(golden_ops[f"sampling/{i}/probs"]))
        assert torch.equal(ops_ref.greedy(logits), t(golden_ops[f"sampling/{i}/argmax"]))
        torch.manual_seed(1234 + i)
        ids = ops_ref.top_k_top_p_min_p_sampling_from_probs(probs, ks, ps, None)
        assert torch.equal(ids, t(golden_ops[f"sampling/{i}/ids_topk_topp"]))
        torch.manual_seed(1234 + i)
        ids = ops_ref.top_k_top_p_min_p_sampling_from_probs(probs, ks, ps, mps)
        assert torch.equal(ids, t(golden_ops[f"sampling/{i}/ids_minp"]))
        # every reference draw lies in the oracle's kept set
        dist = ops_ref.kept_distribution(probs, ks, ps, None)
        assert (dist.gather(1, t(golden_ops[f"sampling/{i}/ids_topk_topp"]).long()[:, None]) > 0).all()


def test_index_ops_match_reference(golden_ops):
    pre = torch.tensor([3, 0, 17], dtype=torch.int32)
    ext = torch.tensor([5, 9, 1], dtype=torch.int32)
    p, s = ops_ref.compute_position(pre, ext)
    assert torch.equal(p, t(golden_ops["compute_position/positions"]))
    assert torch.equal(s, t(golden_ops["compute_position/start"]))
    assert ops_ref.clamp_position(torch.tensor([0, 1, 9])).tolist() == [0, 0, 8]


def test_fp8_fallback_gemm_matches_reference(golden_ops):
    if "fp8/fallback_scaled_mm" not in golden_ops:
        pytest.skip("reference fp8 fallback GEMM was not runnable on CPU when goldens were made")
    x = cases.randn((5, 256), 900, torch.bfloat16)
    w = cases.randn((48, 256), 901, torch.float32, 0.05)
    ws = w.abs().amax(1, keepdim=True) / 448.0
    wq = (w / ws).clamp(-448, 448).to(torch.float8_e4m3fn)
    xq, xs = ops_ref.per_token_quant_fp8(x)
    assert torch.equal(xq.float(), t(golden_ops["fp8/xq"]))
    assert torch.equal(xs, t(golden_ops["fp8/xs"]))
    y = ops_ref.fp8_linear(x, wq, ws)
    assert (y.float() - t(golden_ops["fp8/fallback_scaled_mm"])).abs().max() <= 1e-2


@pytest.mark.parametrize("name,preset,dtn,lens,steps", cases.E2E_CASES)
def test_e2e_matches_reference(name, preset, dtn, lens, steps):
    g = load_e2e(preset, dtn)
    dt = DT[dtn]
    tname = {"f32": "float32", "bf16": "bfloat16"}[dtn]
    cfg = hf_config_dict(preset, tname)
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=dt)
    m = OracleLlama(cfg, w, dt, max_total_tokens=2048)
    prompts = cases.e2e_prompts(cfg["vocab_size"], lens)
    logits, loc = m.extend(prompts)
    assert torch.equal(loc, t(g[f"e2e/{name}/prefill_out_cache_loc"]))
    tol = 1e-5 if dtn == "f32" else 0.0  # bf16: same op sequence on the same build -> same bits
    assert (logits - t(g[f"e2e/{name}/prefill_logits"])).abs().max() <= tol
    toks = [ops_ref.greedy(logits)]
    for s in range(steps - 1):
        logits, loc = m.decode(toks[-1])
        assert torch.equal(loc, t(g[f"e2e/{name}/decode_out_cache_loc"][s]))
        assert (logits - t(g[f"e2e/{name}/decode_logits"][s])).abs().max() <= tol
        toks.append(ops_ref.greedy(logits))
    assert torch.equal(torch.stack(toks, 1), t(g[f"e2e/{name}/tokens"]))
    n = g[f"e2e/{name}/req_to_token"].shape[1]
    assert torch.equal(m.req_to_token[: len(prompts), :n], t(g[f"e2e/{name}/req_to_token"]))
    if f"e2e/{name}/prefix_extend_logits" in g:
        cut = cases.E2E_PREFIX_CUT
        m.clear()
        m.extend([p[:cut] for p in prompts])
        logits, loc = m.extend(prompts, prefix_lens=[cut] * len(prompts))
        assert torch.equal(loc, t(g[f"e2e/{name}/prefix_extend_out_cache_loc"]))
        assert (logits - t(g[f"e2e/{name}/prefix_extend_logits"])).abs().max() <= tol


# ---- DeepSeek multi-head latent attention: tests/golden/mla.npz (gen_mla_golden.py) ------------------
@pytest.fixture(scope="module")
def golden_mla():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mla.npz"))


@pytest.mark.parametrize("dtn", list(DT))
def test_mla_decode_kernel_matches_reference(golden_mla, dtn):
    """Multi-query attention over a 576-wide latent cache whose value is the row's first 512 features:
    the oracle's decode routine == the reference's torch-native routine bit for bit; the reference's
    C++ kernel (MLA branch, decode.cpp:1197) is held to its own test's cosine bar."""
    R, P = cases.MLA_RANK, cases.MLA_ROPE
    for i, (B, Hq, _) in enumerate(cases.MLA_DECODE_CASES):
        q, kv, r2t, pool, lens = cases.mla_decode_inputs(i, DT[dtn])
        o = ops_ref.decode_attention(q, kv, kv[..., :R], r2t, pool, lens, (R + P) ** -0.5, True)
        assert torch.equal(o.float(), t(golden_mla[f"mla_decode/{dtn}/{i}"]))
        for b in range(B):
            toks = r2t[pool[b], : lens[b]].long()
            ref = ops_ref.attention_fp64(q[b][:, None], kv[toks].transpose(0, 1), kv[toks][..., :R].transpose(0, 1),
                                         (R + P) ** -0.5)[:, 0]
            assert (o[b].double() - ref).abs().max() < (1e-5 if dtn == "f32" else 2e-2)
        if dtn == "bf16":
            ck = t(golden_mla[f"mla_decode_cpu_kernel/{dtn}/{i}"])
            assert torch.nn.functional.cosine_similarity(ck.flatten(), o.float().flatten(), dim=0) > 0.99


@pytest.mark.parametrize("name,dtn,qr,scaling", cases.MLA_MODULE_CASES)
def test_mla_module_matches_reference(golden_mla, name, dtn, qr, scaling):
    """The module oracle replays the generator's scenario: absorbed prefill + decode steps, the
    multi-head form of the same prefill, and a two-chunk prefill whose second chunk extends over the
    cached latent rows.  Same torch build, same op sequence -> the reference's bits."""
    from oracle import mla_ref

    dt, d = DT[dtn], cases.MLA_DIMS
    g = lambda k: t(golden_mla[f"mla_module/{name}/{k}"])   # noqa: E731
    w = cases.mla_weights(qr)

    def new():
        return mla_ref.OracleMLA(w, dt, d["hidden"], d["heads"], d["nope"], d["rope"], d["v"], d["kv_lora"], qr,
                                 cases.MLA_THETA, cases.MLA_EPS, scaling)

    m = new()
    assert abs(m.scaling - float(golden_mla[f"mla_module/{name}/softmax_scale"])) < 1e-15
    assert torch.equal(m.cache[:128].float(), g("cos_sin_cache_head"))
    lens, steps, rows = cases.MLA_LENS, cases.MLA_STEPS, torch.tensor([2, 0, 5])
    seqs = cases.mla_hidden_states(lens, steps, 9, dt)
    slots = cases.mla_slots(sum(lens) + steps * len(lens), 21)
    per, p = [], 0
    for n in lens:
        per.append(slots[p: p + n])
        p += n
    zeros, lt = torch.zeros(3, dtype=torch.int64), torch.tensor(lens)
    tol = 0.0 if dtn != "f32" else 1e-6     # fp32 GEMMs may block differently run to run; 16-bit outputs round it away

    def close(a, key):
        assert (a.float() - g(key)).abs().max().item() <= tol * max(1.0, g(key).abs().max().item()), key

    with torch.no_grad():
        for r, sl in zip(rows, per):
            m.req_to_token[r, : len(sl)] = sl.int()
        loc = torch.cat(per)
        pos, _ = ops_ref.compute_position(zeros, lt)
        x = torch.cat([s[:n] for s, n in zip(seqs, lens)])
        close(m.absorb("extend", pos, x, rows, lt, zeros, lt, loc), "absorb_prefill")
        close(m.kv[loc, 0], "latent_cache")
        for step in range(steps):
            new_slots = slots[p: p + len(lens)]
            p += len(lens)
            m.req_to_token[rows, lt + step] = new_slots.int()
            x = torch.stack([s[n + step] for s, n in zip(seqs, lens)])
            sl = lt + step + 1
            close(m.absorb("decode", sl - 1, x, rows, sl, None, None, new_slots), f"decode_{step}")
        # multi-head form of the same prefill
        m = new()
        for r, sl in zip(rows, per):
            m.req_to_token[r, : len(sl)] = sl.int()
        x = torch.cat([s[:n] for s, n in zip(seqs, lens)])
        q, k, v = m.normal_prepare(pos, x, loc)
        close(m.kv[loc, 0], "normal_latent_cache")
        close(m.normal_core(q, k, v, rows, lt, zeros, lt, loc), "normal_prefill")
        # two chunks
        first, clens = torch.tensor(cases.MLA_CHUNK_FIRST), torch.tensor(cases.MLA_CHUNK_LENS)
        cseqs = cases.mla_hidden_states(cases.MLA_CHUNK_LENS, 1, 12, dt)
        m = new()
        slots = cases.mla_slots(int(clens.sum()), 22)
        per1, per2, p = [], [], 0
        for f, n in zip(first.tolist(), clens.tolist()):
            per1.append(slots[p: p + f])
            per2.append(slots[p + f: p + n])
            p += n
        for r, a, b in zip(rows, per1, per2):
            m.req_to_token[r, : len(a)] = a.int()
            m.req_to_token[r, len(a): len(a) + len(b)] = b.int()
        pos1, _ = ops_ref.compute_position(zeros, first)
        x = torch.cat([s[:f] for s, f in zip(cseqs, first.tolist())])
        close(m.absorb("extend", pos1, x, rows, first, zeros, first, torch.cat(per1)), "chunk1")
        pos2, _ = ops_ref.compute_position(first, clens - first)
        x = torch.cat([s[f:n] for s, f, n in zip(cseqs, first.tolist(), clens.tolist())])
        close(m.absorb("extend", pos2, x, rows, clens, first, clens - first, torch.cat(per2)), "chunk2")

    # the float64 functions (expanded and absorbed) agree with the reference's outputs to the dtype's noise
    tol64 = 2e-5 if dtn == "f32" else 4e-2
    wd = {k: v.to(dt) for k, v in w.items()}
    for fn in (mla_ref.mla_attention, mla_ref.mla_attention_absorbed):
        want = [fn(s, wd, d["heads"], d["nope"], d["rope"], d["v"], qr, d["kv_lora"], cases.MLA_THETA, cases.MLA_EPS, scaling)
                for s in seqs]
        ref = torch.cat([r[:n] for r, n in zip(want, lens)]).float()
        for key in ("absorb_prefill", "normal_prefill"):
            assert (ref - g(key)).abs().max().item() <= tol64 * g(key).abs().max().item(), (fn.__name__, key)
        for step in range(steps):
            ref = torch.stack([r[n + step] for r, n in zip(want, lens)]).float()
            assert (ref - g(f"decode_{step}")).abs().max().item() <= tol64 * g("absorb_prefill").abs().max().item()


@pytest.mark.parametrize("name,preset,dtn,lens,steps", cases.E2E_LONG_CASES)
def test_e2e_real_shape_matches_reference(name, preset, dtn, lens, steps):
    """Configuration 1 at its real shape (Llama-3.2-1B, vocab 128256, tied head, llama3 rope scaling):
    one request, 128-token prompt, 128 greedy tokens -- the oracle reproduces the reference's tokens
    and its logits bit for bit (same op sequence on the same torch build)."""
    g = load_e2e(preset, dtn)
    dt = DT[dtn]
    cfg = hf_config_dict(preset, {"f32": "float32", "bf16": "bfloat16"}[dtn])
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=dt)
    m = OracleLlama(cfg, w, dt, max_total_tokens=2048, max_reqs=4, context_len=512)
    prompts = cases.e2e_prompts(10000, lens)
    logits, _ = m.extend(prompts)
    tol = 0.0 if dtn == "bf16" else 1e-5      # fp32 GEMMs may block differently from run to run
    assert (logits - t(g[f"e2e/{name}/prefill_logits"])).abs().max() <= tol
    ref_toks = t(g[f"e2e/{name}/tokens"])
    tv = t(g[f"e2e/{name}/top5_values"])
    if dtn != "bf16":
        steps = 24                              # the fp32 model is the slow one on CPU: its first 24 steps
    toks = [ops_ref.greedy(logits)]
    for s in range(1, steps):
        logits, _ = m.decode(toks[-1])
        v, i = logits.topk(5, dim=-1)
        assert (v - tv[:, s]).abs().max() <= tol, s
        toks.append(ops_ref.greedy(logits))
    if steps == ref_toks.shape[1]:
        assert (logits - t(g[f"e2e/{name}/last_logits"])).abs().max() <= tol
    assert torch.equal(torch.stack(toks, 1), ref_toks[:, :steps])


def test_mla_module_with_fp8_kv_b_proj_matches_reference(golden_mla):
    """kv_b_proj from an fp8 checkpoint: the reference's post_load_weights re-quantises it per tensor and
    the absorbed path multiplies by w.to(bf16) * w_scale; the oracle reproduces the tensor scale and
    the module's prefill / decode outputs bit for bit."""
    from oracle import mla_ref

    name, dtn, qr, scaling = cases.MLA_FP8_CASE
    dt, d = DT[dtn], cases.MLA_DIMS
    g = lambda k: t(golden_mla[f"mla_module/{name}/{k}"])   # noqa: E731
    w = cases.mla_weights(qr)
    m = mla_ref.OracleMLA(w, dt, d["hidden"], d["heads"], d["nope"], d["rope"], d["v"], d["kv_lora"], qr, cases.MLA_THETA,
                          cases.MLA_EPS, scaling, kv_b_fp8=cases.mla_kv_b_fp8(w["kv_b_proj"]))
    assert abs(float(m.w_scale) - float(golden_mla[f"mla_module/{name}/w_scale"])) < 1e-12
    lens, steps, rows = cases.MLA_LENS, cases.MLA_STEPS, torch.tensor([2, 0, 5])
    seqs = cases.mla_hidden_states(lens, steps, 9, dt)
    slots = cases.mla_slots(sum(lens) + steps * len(lens), 21)
    per, p = [], 0
    for n in lens:
        per.append(slots[p: p + n])
        p += n
    zeros, lt = torch.zeros(3, dtype=torch.int64), torch.tensor(lens)
    with torch.no_grad():
        for r, sl in zip(rows, per):
            m.req_to_token[r, : len(sl)] = sl.int()
        loc = torch.cat(per)
        pos, _ = ops_ref.compute_position(zeros, lt)
        x = torch.cat([s[:n] for s, n in zip(seqs, lens)])
        assert torch.equal(m.absorb("extend", pos, x, rows, lt, zeros, lt, loc).float(), g("absorb_prefill"))
        for step in range(steps):
            new_slots = slots[p: p + len(lens)]
            p += len(lens)
            m.req_to_token[rows, lt + step] = new_slots.int()
            x = torch.stack([s[n + step] for s, n in zip(seqs, lens)])
            sl = lt + step + 1
            assert torch.equal(m.absorb("decode", sl - 1, x, rows, sl, None, None, new_slots).float(), g(f"decode_{step}"))
