"""Generate tests/golden/mla.npz by running the REFERENCE's DeepSeek MLA code on the inputs of
cases.py (build container only; the reference tree lives at /root/reference there):
    python tests/golden/gen_mla_golden.py

What runs is the reference's own code: DeepseekV2AttentionMLA (models/deepseek_v2.py:746-1335) with its
linears, norms, rotary embedding (incl. DeepseekScalingRotaryEmbedding = YaRN), MLATokenToKVPool,
DeepseekV2ForCausalLM.post_load_weights (w_kc / w_vc) and TorchNativeAttnBackend; and, for the
attention kernel alone, TorchNativeAttnBackend._run_sdpa_forward_decode and the native C++
decode_attention_cpu on a 576-wide latent cache.  The fixture holds outputs only.

Two facts about the reference on CPU that shape this script:
 * its torch-native backend reads keys from the pool, so the multi-head ("normal") prefill of the
   module cannot run through it on an MLA pool; the module's forward_normal_prepare is run as is
   (it yields q, k, v and writes the latent cache), and the attention itself goes through the same
   backend's SDPA routine on a multi-head-shaped pool holding exactly that k and v;
 * the absorbed form runs through the backend unchanged, for prefill, prefix-extend and decode.
"""
import os
import sys
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import gen_golden  # noqa: E402  (installs the import harness for the reference)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases  # noqa: E402

f32 = gen_golden.f32


def gen_decode_kernel(out):
    from sglang.srt.layers.attention.torch_native_backend import TorchNativeAttnBackend

    be = TorchNativeAttnBackend.__new__(TorchNativeAttnBackend)
    R, P = cases.MLA_RANK, cases.MLA_ROPE
    for name, dt in cases.DTYPES.items():
        for i, (B, Hq, _) in enumerate(cases.MLA_DECODE_CASES):
            q, kv, r2t, pool, lens = cases.mla_decode_inputs(i, dt)
            o = torch.empty((B, Hq, R), dtype=dt)
            be._run_sdpa_forward_decode(q, o, kv, kv[..., :R], r2t, pool, lens, scaling=(R + P) ** -0.5,
                                        enable_gqa=True, causal=False)
            out[f"mla_decode/{name}/{i}"] = f32(o)
            if name == "bf16":
                o2 = torch.empty((B, Hq, R), dtype=dt)
                logits = torch.empty((B, Hq, 8, R + 1), dtype=torch.float32)
                # key/value/loc of the "new token": rewrite an existing row with its own content
                last = torch.stack([r2t[pool[b], lens[b] - 1] for b in range(B)]).to(torch.int64)
                torch.ops.sgl_kernel.decode_attention_cpu(
                    q, kv, kv[..., :R], o2, kv[last].clone(), kv[last][..., :R].clone(), last, logits, r2t, pool,
                    lens, (R + P) ** -0.5, 0.0)
                out[f"mla_decode_cpu_kernel/{name}/{i}"] = f32(o2)


def build_module(dt, q_lora_rank, rope_scaling, fp8_kv_b=False):
    from sglang.srt.managers.schedule_batch import global_server_args_dict
    from sglang.srt.models.deepseek_v2 import DeepseekV2AttentionMLA, DeepseekV2ForCausalLM

    d = cases.MLA_DIMS
    # the module asks get_rope_wrapper for a rotary embedding on global_server_args_dict["device"]; the
    # reference's CPU factory only builds the YaRN class, its general factory builds the plain one
    global_server_args_dict["device"] = "cpu" if rope_scaling else "cuda"
    torch.set_default_dtype(dt)
    try:
        m = DeepseekV2AttentionMLA(
            config=SimpleNamespace(rms_norm_eps=cases.MLA_EPS), hidden_size=d["hidden"], num_heads=d["heads"],
            qk_nope_head_dim=d["nope"], qk_rope_head_dim=d["rope"], v_head_dim=d["v"], q_lora_rank=q_lora_rank,
            kv_lora_rank=d["kv_lora"], rope_theta=cases.MLA_THETA,
            rope_scaling=dict(rope_scaling) if rope_scaling else None,
            max_position_embeddings=4096, quant_config=None, layer_id=0, prefix="attn")
    finally:
        torch.set_default_dtype(torch.float32)
        global_server_args_dict["device"] = "cpu"
    w = {k: v.to(dt) for k, v in cases.mla_weights(q_lora_rank).items()}
    with torch.no_grad():
        for name, val in w.items():
            getattr(m, name).weight.data = val.clone()
    fake = SimpleNamespace(config=SimpleNamespace(num_hidden_layers=1), quant_config=None,
                           model=SimpleNamespace(layers=[SimpleNamespace(self_attn=m)]))
    if fp8_kv_b:
        # an fp8 checkpoint: kv_b_proj.weight e4m3fn + per-channel weight_scale.  post_load_weights
        # re-quantises per tensor (channel_quant_to_tensor_quant) and, on a host without an fp8 bmm,
        # widens w_kc / w_vc to bf16 times the tensor scale (:2291-2297) -- the same expression its ROCm
        # branch evaluates on every call (:1242-1247, 1300-1306).  That widening is keyed on the
        # module-level CPU flags, set for the duration of this call.
        import sglang.srt.models.deepseek_v2 as ds

        wq, ws = cases.mla_kv_b_fp8(cases.mla_weights(q_lora_rank)["kv_b_proj"])
        m.kv_b_proj.weight.data = wq
        m.kv_b_proj.weight_scale = torch.nn.Parameter(ws, requires_grad=False)
        saved = (ds._is_cpu, ds._is_cpu_amx_available)
        ds._is_cpu, ds._is_cpu_amx_available = True, True
        try:
            DeepseekV2ForCausalLM.post_load_weights(fake)
        finally:
            ds._is_cpu, ds._is_cpu_amx_available = saved
        assert m.w_kc.dtype == torch.bfloat16
    else:
        DeepseekV2ForCausalLM.post_load_weights(fake)
    return m


class World:
    """Pools + a torch-native backend + hand-made ForwardBatches (the reference's own classes)."""

    def __init__(self, dt, slots=512):
        from sglang.srt.layers.attention.torch_native_backend import TorchNativeAttnBackend
        from sglang.srt.mem_cache.memory_pool import MHATokenToKVPool, MLATokenToKVPool, ReqToTokenPool

        d = cases.MLA_DIMS
        self.dt = dt
        self.r2t = ReqToTokenPool(8, 256, "cpu", False)
        self.mla_pool = MLATokenToKVPool(slots, 1, dt, d["kv_lora"], d["rope"], 1, "cpu", False)
        self.mha_pool = MHATokenToKVPool(slots, 1, dt, d["heads"], d["nope"] + d["rope"], 1, "cpu", False)
        self.backend = TorchNativeAttnBackend(SimpleNamespace(device="cpu"))

    def batch(self, mode, pool_rows, prefix, ext, loc, pool=None):
        from sglang.srt.model_executor.forward_batch_info import ForwardBatch, ForwardMode

        B = len(pool_rows)
        seq = [p + e for p, e in zip(prefix, ext)]
        if mode == "decode":
            positions = torch.tensor([s - 1 for s in seq], dtype=torch.int64)
        else:
            positions = torch.cat([torch.arange(p, p + e) for p, e in zip(prefix, ext)]).to(torch.int64)
        return ForwardBatch(
            forward_mode=ForwardMode.DECODE if mode == "decode" else ForwardMode.EXTEND, batch_size=B,
            input_ids=torch.zeros(sum(ext), dtype=torch.int64), req_pool_indices=torch.tensor(pool_rows, dtype=torch.int64),
            seq_lens=torch.tensor(seq, dtype=torch.int64), out_cache_loc=loc.to(torch.int64), seq_lens_sum=sum(seq),
            positions=positions, extend_num_tokens=sum(ext), extend_seq_lens=torch.tensor(ext, dtype=torch.int32),
            extend_prefix_lens=torch.tensor(prefix, dtype=torch.int32), extend_prefix_lens_cpu=list(prefix),
            extend_seq_lens_cpu=list(ext), req_to_token_pool=self.r2t, token_to_kv_pool=pool or self.mla_pool,
            attn_backend=self.backend)


def write_table(world, rows, starts, slot_lists):
    for r, s, sl in zip(rows, starts, slot_lists):
        world.r2t.req_to_token[r, s: s + len(sl)] = sl.to(torch.int32)


def gen_module(out):
    from sglang.srt.model_executor.forward_batch_info import ForwardMode  # noqa: F401

    d = cases.MLA_DIMS
    for name, dtn, qr, scaling in cases.MLA_MODULE_CASES + [cases.MLA_FP8_CASE]:
        dt = cases.DTYPES[dtn]
        fp8 = name == cases.MLA_FP8_CASE[0]
        m = build_module(dt, qr, scaling, fp8_kv_b=fp8)
        if fp8:
            out[f"mla_module/{name}/w_scale"] = np.float64(float(m.w_scale))
        out[f"mla_module/{name}/softmax_scale"] = np.float64(m.scaling)
        out[f"mla_module/{name}/cos_sin_cache_head"] = f32(m.rotary_emb.cos_sin_cache[:128])
        lens, steps = cases.MLA_LENS, cases.MLA_STEPS
        seqs = cases.mla_hidden_states(lens, steps, 9, dt)
        rows = [2, 0, 5]
        with torch.no_grad():
            # ---- (1) absorbed form: prefill, then decode steps --------------------------------------
            w = World(dt)
            slots = cases.mla_slots(sum(lens) + steps * len(lens), 21)
            per, p = [], 0
            for n in lens:
                per.append(slots[p: p + n])
                p += n
            write_table(w, rows, [0] * len(lens), per)
            fb = w.batch("extend", rows, [0] * len(lens), list(lens), torch.cat(per))
            x = torch.cat([s[:n] for s, n in zip(seqs, lens)])
            o = m.forward_absorb_core(*m.forward_absorb_prepare(fb.positions, x.clone(), fb, None))
            out[f"mla_module/{name}/absorb_prefill"] = f32(o)
            loc_all = torch.cat(per)
            out[f"mla_module/{name}/latent_cache"] = f32(w.mla_pool.get_key_buffer(0)[loc_all, 0])
            for t in range(steps):
                new = slots[p: p + len(lens)]
                p += len(lens)
                write_table(w, rows, [n + t for n in lens], [new[i: i + 1] for i in range(len(lens))])
                fb = w.batch("decode", rows, [n + t for n in lens], [1] * len(lens), new)
                x = torch.stack([s[n + t] for s, n in zip(seqs, lens)])
                o = m(fb.positions, x.clone(), fb, None)       # dispatch -> absorbed form on decode
                out[f"mla_module/{name}/decode_{t}"] = f32(o)

            if fp8:
                continue   # the multi-head prefill would run kv_b_proj as an fp8 W8A8 linear: not this scenario
            # ---- (2) multi-head form of the same prefill ---------------------------------------------
            w = World(dt)
            write_table(w, rows, [0] * len(lens), per)
            fb = w.batch("extend", rows, [0] * len(lens), list(lens), torch.cat(per))
            x = torch.cat([s[:n] for s, n in zip(seqs, lens)])
            q, k, v, _ = m.forward_normal_prepare(fb.positions, x.clone(), fb, None)
            out[f"mla_module/{name}/normal_latent_cache"] = f32(w.mla_pool.get_key_buffer(0)[loc_all, 0])
            # attention over exactly this k / v: the backend's SDPA routine reads a multi-head pool
            kb, vb = w.mha_pool.get_key_buffer(0), w.mha_pool.get_value_buffer(0)
            kb[loc_all] = k
            vb[loc_all, :, : d["v"]] = v
            H = d["heads"]
            o_ = torch.empty((q.shape[0], H, d["v"]), dtype=dt)
            w.backend._run_sdpa_forward_extend(q, o_, kb, vb[..., : d["v"]], w.r2t.req_to_token, fb.req_pool_indices,
                                               fb.seq_lens, fb.extend_prefix_lens, fb.extend_seq_lens,
                                               scaling=m.attn_mha.scaling, enable_gqa=False, causal=True)
            o, _ = m.o_proj(o_.reshape(-1, H * d["v"]))
            out[f"mla_module/{name}/normal_prefill"] = f32(o)

            # ---- (3) chunked prefill: chunk 2 extends over the cached latent rows (absorbed form) ----
            first, clens = cases.MLA_CHUNK_FIRST, cases.MLA_CHUNK_LENS
            cseqs = cases.mla_hidden_states(clens, 1, 12, dt)
            w = World(dt)
            slots = cases.mla_slots(sum(clens), 22)
            per1, per2, p = [], [], 0
            for f, n in zip(first, clens):
                per1.append(slots[p: p + f])
                per2.append(slots[p + f: p + n])
                p += n
            write_table(w, rows, [0] * 3, per1)
            fb = w.batch("extend", rows, [0] * 3, list(first), torch.cat(per1))
            x = torch.cat([s[:f] for s, f in zip(cseqs, first)])
            o = m.forward_absorb_core(*m.forward_absorb_prepare(fb.positions, x.clone(), fb, None))
            out[f"mla_module/{name}/chunk1"] = f32(o)
            write_table(w, rows, list(first), per2)
            fb = w.batch("extend", rows, list(first), [n - f for f, n in zip(first, clens)], torch.cat(per2))
            x = torch.cat([s[f:n] for s, f, n in zip(cseqs, first, clens)])
            o = m(fb.positions, x.clone(), fb, None)           # dispatch -> absorbed form (prefix > 0)
            out[f"mla_module/{name}/chunk2"] = f32(o)


def main():
    # one tiny reference ModelRunner first: it initialises the reference's process-global state
    # (distributed groups of size 1, attention-TP info, global_server_args_dict)
    gen_golden.load_reference_runner("tiny-llama-d64", "f32")
    out = {}
    gen_decode_kernel(out)
    gen_module(out)
    path = os.path.join(HERE, "mla.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
