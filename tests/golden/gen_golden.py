"""Generate tests/golden/*.npz by running the REFERENCE's own code (python/sglang/srt and the
sgl-kernel CPU kernels compiled into oracle/_ref/) on the inputs of cases.py.

Run in the build container only (the reference tree lives at /root/reference there):
    python tests/golden/gen_golden.py
The fixtures hold outputs only (plus tiny integer inputs); no reference source or bytecode is
written anywhere.  bf16/f16 arrays are stored as float32 (exactly representable).
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import _ref_harness  # noqa: E402

_ref_harness.install()
_ref_harness.use_amx_kernels(False)   # canonical oracle path: torch-native (forward_native, F.linear)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases  # noqa: E402
from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights, LLAMA_PRESETS  # noqa: E402

from sglang.srt.layers.activation import SiluAndMul  # noqa: E402
from sglang.srt.layers.attention.torch_native_backend import TorchNativeAttnBackend  # noqa: E402
from sglang.srt.layers.layernorm import RMSNorm  # noqa: E402
from sglang.srt.layers.rotary_embedding import get_rope  # noqa: E402
from sglang.srt.layers.sampler import top_k_top_p_min_p_sampling_from_probs_torch  # noqa: E402
from sglang.srt.model_executor.forward_batch_info import compute_position_torch  # noqa: E402


def f32(t):
    return t.detach().to(torch.float32).numpy()


def gen_ops(out):
    for name, dt in cases.DTYPES.items():
        torch.set_default_dtype(dt)
        for i, (rows, h) in enumerate(cases.NORM_CASES):
            x, r, w = cases.norm_inputs(i, dt)
            m = RMSNorm(h, eps=1e-5)
            m.weight.data = w.clone()
            out[f"rmsnorm/{name}/{i}"] = f32(m.forward_native(x.clone()))
            y, r2 = m.forward_native(x.clone(), r.clone())
            out[f"fused_add_rmsnorm/{name}/{i}/x"] = f32(y)
            out[f"fused_add_rmsnorm/{name}/{i}/residual"] = f32(r2)
        for i in range(len(cases.ACT_CASES)):
            out[f"silu_and_mul/{name}/{i}"] = f32(SiluAndMul().forward_native(cases.act_inputs(i, dt)))
        for i, (t, hq, hkv, hd, rd, neox, l3) in enumerate(cases.ROPE_CASES):
            pos, q, k = cases.rope_inputs(i, dt)
            rope = get_rope(hd, rotary_dim=rd, max_position=2048, base=500000.0 if l3 else 10000.0,
                            is_neox_style=neox, rope_scaling=dict(cases.LLAMA3) if l3 else None, dtype=dt)
            q2, k2 = rope.forward_native(pos, q.clone(), k.clone())
            out[f"rope/{name}/{i}/q"] = f32(q2)
            out[f"rope/{name}/{i}/k"] = f32(k2)
            if name == "f32":
                out[f"rope_cache/{i}"] = f32(rope.cos_sin_cache[::37])  # every 37th position
        be = TorchNativeAttnBackend.__new__(TorchNativeAttnBackend)
        for i, (B, Hq, Hkv, D, _) in enumerate(cases.DECODE_CASES):
            q, kc, vc, r2t, pool, lens = cases.decode_inputs(i, dt)
            o = torch.empty_like(q)
            be._run_sdpa_forward_decode(q, o, kc, vc, r2t, pool, lens, scaling=D ** -0.5,
                                        enable_gqa=Hq != Hkv, causal=False)
            out[f"decode/{name}/{i}"] = f32(o)
            if name in ("bf16", "f16"):
                # the reference's native C++ CPU kernel on the same inputs (sgl-kernel/csrc/cpu/decode.cpp)
                o2 = torch.empty_like(q)
                logits = torch.empty((B, Hq, 8, D + 1), dtype=torch.float32)
                torch.ops.sgl_kernel.decode_attention_cpu(
                    q, kc, vc, o2, torch.zeros(B, Hkv, D, dtype=dt), torch.zeros(B, Hkv, D, dtype=dt),
                    torch.zeros(B, dtype=torch.int64), logits, r2t, pool, lens, D ** -0.5, 0.0)
                out[f"decode_cpu_kernel/{name}/{i}"] = f32(o2)
        for i, (Hq, Hkv, D, pre, ext) in enumerate(cases.EXTEND_CASES):
            q, kc, vc, r2t, pool, seq, pre_t, ext_t = cases.extend_inputs(i, dt)
            o = torch.empty_like(q)
            be._run_sdpa_forward_extend(q, o, kc, vc, r2t, pool, seq, pre_t, ext_t, scaling=D ** -0.5,
                                        enable_gqa=Hq != Hkv, causal=True)
            out[f"extend/{name}/{i}"] = f32(o)
    torch.set_default_dtype(torch.float32)
    for i in range(len(cases.SAMPLING_CASES)):
        logits, temps, ks, ps, mps = cases.sampling_inputs(i)
        probs = torch.softmax(logits / temps, dim=-1)
        out[f"sampling/{i}/probs"] = f32(probs)
        torch.manual_seed(1234 + i)
        out[f"sampling/{i}/ids_topk_topp"] = top_k_top_p_min_p_sampling_from_probs_torch(
            probs.clone(), ks, ps, mps, False).numpy()
        torch.manual_seed(1234 + i)
        out[f"sampling/{i}/ids_minp"] = top_k_top_p_min_p_sampling_from_probs_torch(
            probs.clone(), ks, ps, mps, True).numpy()
        out[f"sampling/{i}/argmax"] = torch.argmax(logits, -1).numpy()
    pre = torch.tensor([3, 0, 17], dtype=torch.int32)
    ext = torch.tensor([5, 9, 1], dtype=torch.int32)
    p, s = compute_position_torch(pre, ext)
    out["compute_position/positions"] = p.numpy()
    out["compute_position/start"] = s.numpy()


def gen_fp8(out):
    """fp8: the reference's activation quantiser is Triton-only (fp8_kernel.py:115-155) and does
    not run on CPU; what CAN be pinned is its unfused dequant GEMM (_apply_fallback_scaled_mm)."""
    from sglang.srt.layers.quantization import fp8_utils

    x = cases.randn((5, 256), 900, torch.bfloat16)
    w = cases.randn((48, 256), 901, torch.float32, 0.05)
    ws = w.abs().amax(1, keepdim=True) / 448.0
    wq = (w / ws).clamp(-448, 448).to(torch.float8_e4m3fn)
    xs = x.float().abs().amax(1, keepdim=True).clamp(min=1e-10) / 448.0
    xq = (x.float() * (1.0 / xs)).clamp(-448, 448).to(torch.float8_e4m3fn)
    try:
        y = fp8_utils._apply_fallback_scaled_mm(xq, wq.t(), xs, ws, x.shape, [5, 48], None, x.dtype)
        out["fp8/fallback_scaled_mm"] = f32(y)
        out["fp8/xq"] = f32(xq)
        out["fp8/xs"] = f32(xs)
    except Exception as e:  # torch._scaled_mm may be unavailable on CPU
        print("fp8 fallback scaled mm not runnable on CPU here:", repr(e))


def load_reference_runner(preset, dtype_name, context_length=None):
    from safetensors.torch import save_file
    from sglang.srt.configs.model_config import ModelConfig
    from sglang.srt.model_executor.model_runner import ModelRunner
    from sglang.srt.server_args import PortArgs, ServerArgs

    dt = cases.DTYPES[dtype_name]
    tname = {"f32": "float32", "bf16": "bfloat16", "f16": "float16"}[dtype_name]
    d = tempfile.mkdtemp()
    cfg = hf_config_dict(preset, tname)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=dt)
    save_file({k: v.contiguous() for k, v in w.items()}, os.path.join(d, "model.safetensors"))
    sa = ServerArgs(model_path=d, load_format="auto", device="cpu", attention_backend="torch_native",
                    sampling_backend="pytorch", disable_cuda_graph=True, skip_tokenizer_init=True,
                    dtype=tname, max_total_tokens=2048, disable_radix_cache=True,
                    chunked_prefill_size=-1, mem_fraction_static=0.5, context_length=context_length)
    pa = PortArgs.init_new(sa)
    mc = ModelConfig.from_server_args(sa)
    # The container's transformers (5.x) moves rope_theta into rope_parameters; the reference
    # is pinned to 4.53.2 where config.rope_theta exists (models/llama.py:204).  Present it so.
    if getattr(mc.hf_config, "rope_theta", None) is None:
        mc.hf_config.rope_theta = cfg["rope_theta"]
    mr = ModelRunner(model_config=mc, mem_fraction_static=0.5, gpu_id=0, tp_rank=0, tp_size=1,
                     pp_rank=0, pp_size=1, nccl_port=pa.nccl_port, server_args=sa)
    return mr, cfg


_RUNNERS = {}


def gen_e2e(out):
    from sglang.srt.managers.schedule_batch import Req, ScheduleBatch
    from sglang.srt.model_executor.forward_batch_info import ForwardBatch
    from sglang.srt.sampling.sampling_params import SamplingParams
    from sglang.srt.speculative.spec_info import SpeculativeAlgorithm

    def new_batch(reqs, mr):
        return ScheduleBatch.init_new(
            reqs=reqs, req_to_token_pool=mr.req_to_token_pool,
            token_to_kv_pool_allocator=mr.token_to_kv_pool_allocator, tree_cache=None,
            model_config=mr.model_config, enable_overlap=False,
            spec_algorithm=SpeculativeAlgorithm.NONE, enable_custom_logit_processor=False)

    def mk_req(i, ids):
        r = Req(rid=i, origin_input_text="", origin_input_ids=list(ids),
                sampling_params=SamplingParams(temperature=0, max_new_tokens=16))
        r.prefix_indices = []
        r.fill_ids = r.origin_input_ids
        r.extend_input_len = len(r.fill_ids)
        r.logprob_start_len = len(r.origin_input_ids) - 1
        return r

    def run(batch, mr):
        fb = ForwardBatch.init_new(batch.get_model_worker_batch(), mr)
        lo, _ = mr.forward(fb)
        return mr.sample(lo, fb), lo.next_token_logits.clone(), fb

    for name, preset, dtn, lens, steps in cases.E2E_CASES:
        if (preset, dtn) not in _RUNNERS:
            continue  # one reference model per process (global parallel state): see main()
        mr, cfg = _RUNNERS[(preset, dtn)]
        prompts = cases.e2e_prompts(cfg["vocab_size"], lens)
        with torch.no_grad():
            mr.req_to_token_pool.clear()
            mr.token_to_kv_pool_allocator.clear()
            batch = new_batch([mk_req(i, p) for i, p in enumerate(prompts)], mr)
            batch.prepare_for_extend()
            nxt, logits, fb = run(batch, mr)
            out[f"e2e/{name}/prefill_logits"] = f32(logits)
            out[f"e2e/{name}/prefill_out_cache_loc"] = batch.out_cache_loc.numpy().copy()
            out[f"e2e/{name}/prefill_positions"] = fb.positions.numpy().copy()
            toks, step_logits, locs = [nxt.clone()], [], []
            for _ in range(steps - 1):
                batch.output_ids = nxt
                batch.prepare_for_decode()
                nxt, logits, fb = run(batch, mr)
                toks.append(nxt.clone())
                step_logits.append(f32(logits))
                locs.append(batch.out_cache_loc.numpy().copy())
            out[f"e2e/{name}/tokens"] = torch.stack(toks, 1).numpy()
            out[f"e2e/{name}/decode_logits"] = np.stack(step_logits)
            out[f"e2e/{name}/decode_out_cache_loc"] = np.stack(locs)
            out[f"e2e/{name}/req_to_token"] = mr.req_to_token_pool.req_to_token[: len(prompts), : max(lens) + steps].numpy().copy()

            # scenario 2: prefix in cache (bench_one_batch.py:197-208): cache prompt[:cut], then
            # extend with the remainder reading the prefix through req_to_token
            cut = cases.E2E_PREFIX_CUT
            if min(lens) > cut:
                mr.req_to_token_pool.clear()
                mr.token_to_kv_pool_allocator.clear()
                reqs = [mk_req(i, p[:cut]) for i, p in enumerate(prompts)]
                batch = new_batch(reqs, mr)
                batch.prepare_for_extend()
                run(batch, mr)
                for i, r in enumerate(reqs):
                    r.fill_ids = r.fill_ids + prompts[i][cut:]
                    r.origin_input_ids = list(r.fill_ids)
                    r.prefix_indices = mr.req_to_token_pool.req_to_token[i, :cut]
                    r.extend_input_len = len(r.fill_ids) - len(r.prefix_indices)
                    r.logprob_start_len = len(r.origin_input_ids) - 1
                mr.req_to_token_pool.clear()   # the same request slots are handed out again
                batch = new_batch(reqs, mr)
                batch.prepare_for_extend()
                nxt, logits, fb = run(batch, mr)
                out[f"e2e/{name}/prefix_extend_logits"] = f32(logits)
                out[f"e2e/{name}/prefix_extend_out_cache_loc"] = batch.out_cache_loc.numpy().copy()
                out[f"e2e/{name}/prefix_extend_positions"] = fb.positions.numpy().copy()


def gen_e2e_long(out):
    """One long greedy run at a real model shape: compact record (see cases.E2E_LONG_CASES)."""
    from sglang.srt.managers.schedule_batch import Req, ScheduleBatch
    from sglang.srt.model_executor.forward_batch_info import ForwardBatch
    from sglang.srt.sampling.sampling_params import SamplingParams
    from sglang.srt.speculative.spec_info import SpeculativeAlgorithm

    for name, preset, dtn, lens, steps in cases.E2E_LONG_CASES:
        if (preset, dtn) not in _RUNNERS:
            continue
        mr, cfg = _RUNNERS[(preset, dtn)]
        prompts = cases.e2e_prompts(10000, lens)     # bench_one_batch draws ids below 10000 (bench_one_batch.py:211-232)
        reqs = []
        for i, ids in enumerate(prompts):
            r = Req(rid=i, origin_input_text="", origin_input_ids=list(ids),
                    sampling_params=SamplingParams(temperature=0, max_new_tokens=steps))
            r.prefix_indices = []
            r.fill_ids = r.origin_input_ids
            r.extend_input_len = len(r.fill_ids)
            r.logprob_start_len = len(r.origin_input_ids) - 1
            reqs.append(r)

        def run(batch):
            fb = ForwardBatch.init_new(batch.get_model_worker_batch(), mr)
            lo, _ = mr.forward(fb)
            return mr.sample(lo, fb), lo.next_token_logits.clone()

        with torch.no_grad():
            mr.req_to_token_pool.clear()
            mr.token_to_kv_pool_allocator.clear()
            batch = ScheduleBatch.init_new(
                reqs=reqs, req_to_token_pool=mr.req_to_token_pool,
                token_to_kv_pool_allocator=mr.token_to_kv_pool_allocator, tree_cache=None,
                model_config=mr.model_config, enable_overlap=False,
                spec_algorithm=SpeculativeAlgorithm.NONE, enable_custom_logit_processor=False)
            batch.prepare_for_extend()
            nxt, logits = run(batch)
            out[f"e2e/{name}/prefill_logits"] = f32(logits)
            toks, top_v, top_i = [nxt.clone()], [], []
            tv, ti = logits.float().topk(5, dim=-1)
            top_v.append(f32(tv)); top_i.append(ti.numpy())
            for _ in range(steps - 1):
                batch.output_ids = nxt
                batch.prepare_for_decode()
                nxt, logits = run(batch)
                toks.append(nxt.clone())
                tv, ti = logits.float().topk(5, dim=-1)
                top_v.append(f32(tv)); top_i.append(ti.numpy())
            out[f"e2e/{name}/tokens"] = torch.stack(toks, 1).numpy()
            out[f"e2e/{name}/last_logits"] = f32(logits)
            out[f"e2e/{name}/top5_values"] = np.stack(top_v, 1)      # [B, steps, 5]
            out[f"e2e/{name}/top5_ids"] = np.stack(top_i, 1)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "ops"
    out = {}
    if which == "ops":
        gen_ops(out)
        gen_fp8(out)
        path = os.path.join(HERE, "ops.npz")
    else:
        # `python gen_golden.py e2e <preset> <dtype>`: one model per process
        preset, dtn = sys.argv[2], sys.argv[3]
        long_run = any(c[1] == preset and c[2] == dtn for c in cases.E2E_LONG_CASES)
        _RUNNERS[(preset, dtn)] = load_reference_runner(preset, dtn, context_length=512 if long_run else None)
        gen_e2e(out)
        gen_e2e_long(out)
        path = os.path.join(HERE, f"e2e_{preset}_{dtn}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
