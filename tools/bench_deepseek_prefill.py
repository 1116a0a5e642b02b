"""Prefill of ONE tensor-parallel rank (of 8) of DeepSeek-V3's dense part (see bench_deepseek_shard.py): a batch of
prompts through the multi-head form of MLA (latent expanded by kv_b_proj, 16 heads of 192 / 128), latent rows
written to the compressed cache.  usage: python tools/bench_deepseek_prefill.py [--batch-size 4] [--input-len 4096]
[--quantization w8a8_fp8] [--kv-cache-dtype fp8_e4m3]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--quantization", default=None)
ap.add_argument("--kv-cache-dtype", default="auto")
ap.add_argument("--batch-size", type=int, default=4)
ap.add_argument("--input-len", type=int, default=4096)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
from sglang_custom_amd import bench_one_batch as bob
from sglang_custom_amd.configs.model_config import ModelConfig
from sglang_custom_amd.managers.schedule_batch import Req, ScheduleBatch
from sglang_custom_amd.model_executor.model_runner import ModelRunner
from sglang_custom_amd.model_loader.synthetic import hf_config_dict
from sglang_custom_amd.server_args import ServerArgs
TP, bs, n = 8, args.batch_size, args.input_len
cfg = hf_config_dict("deepseek-v3-dense4", "bfloat16")
sa = ServerArgs(dtype="bfloat16", max_total_tokens=bs * (n + 16) + 64, max_running_requests=bs + 8, context_length=n + 32,
                cuda_graph_bs=[], disable_cuda_graph=True, tp_size=TP, simulate_tp_shard=True, quantization=args.quantization,
                kv_cache_dtype=args.kv_cache_dtype)
mr = ModelRunner(ModelConfig(cfg, dtype="bfloat16", context_length=n + 32), sa, tp_rank=0, tp_size=TP)
times = []
for rep in range(args.reps + 1):
    reqs = [Req(i, [1 + (i + j) % 1000 for j in range(n)]) for i in range(bs)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nxt, _, batch = bob.extend(reqs, mr)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
    bob.clear_pools(mr)
L = cfg["num_hidden_layers"]
print(json.dumps({"workload": f"DeepSeek-V3 dense part, one rank of TP=8, {L} layers, prefill {bs} x {n} tokens", "ms": [round(t, 2) for t in times],
                  "ms_per_layer_best": round(min(times[1:]) / L, 3), "tokens": bs * n}), flush=True)
