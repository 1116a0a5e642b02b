"""Decode step of ONE tensor-parallel rank (of 8) of DeepSeek-V3's dense part at its real shapes: 16 of 128
heads, hidden 7168, q_lora 1536, latent 512 + 64 (MLA, compressed KV cache), dense MLP 18432 / 8, four layers,
batch 64, context 4096, HIP-graph decode, collectives stubbed (a per-GPU kernel-time diagnostic like
bench.py --simulate-tp, not a job-level number).  The cache is filled by allocation only (no prefill: the
contents do not matter to the timing).
usage: python tools/bench_deepseek_shard.py [--quantization w8a8_fp8] [--kv-cache-dtype fp8_e4m3] [--fused-rope off] [out.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--quantization", default=None)
ap.add_argument("--kv-cache-dtype", default="auto")
ap.add_argument("--fused-rope", default="on")
ap.add_argument("--batch-size", type=int, default=64)
ap.add_argument("--context", type=int, default=4096)
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("out", nargs="?")
args = ap.parse_args()
os.environ["MI355_MLA_FUSED_ROPE"] = args.fused_rope

from sglang_custom_amd import bench_one_batch as bob
from sglang_custom_amd.configs.model_config import ModelConfig
from sglang_custom_amd.managers.schedule_batch import Req, ScheduleBatch
from sglang_custom_amd.model_executor.model_runner import ModelRunner
from sglang_custom_amd.model_loader.synthetic import hf_config_dict
from sglang_custom_amd.server_args import ServerArgs

TP, bs, ctx, steps, warm = 8, args.batch_size, args.context, args.steps, 6
cfg = hf_config_dict("deepseek-v3-dense4", "bfloat16")
sa = ServerArgs(dtype="bfloat16", max_total_tokens=bs * (ctx + steps + warm + 8) + 64, max_running_requests=bs + 8,
                context_length=ctx + steps + warm + 16, cuda_graph_bs=[bs], tp_size=TP, simulate_tp_shard=True,
                quantization=args.quantization, kv_cache_dtype=args.kv_cache_dtype)
mr = ModelRunner(ModelConfig(cfg, dtype="bfloat16", context_length=ctx + steps + warm + 16), sa, tp_rank=0, tp_size=TP)
reqs = [Req(i, [1 + (i + j) % 1000 for j in range(ctx)]) for i in range(bs)]
batch = ScheduleBatch.init_new(reqs=reqs, req_to_token_pool=mr.req_to_token_pool,
                               token_to_kv_pool_allocator=mr.token_to_kv_pool_allocator, tree_cache=None,
                               model_config=mr.model_config)
batch.prepare_for_extend()      # slots + page table only: no prefill is run
nxt = torch.ones(bs, dtype=torch.int64, device="cuda")
for _ in range(warm):
    nxt, _ = bob.decode(nxt, batch, mr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    nxt, _ = bob.decode(nxt, batch, mr)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / steps
L = cfg["num_hidden_layers"]
H = cfg["num_attention_heads"] // TP
h, qr, kr, rope, nope, vd, inter = (cfg["hidden_size"], cfg["q_lora_rank"], cfg["kv_lora_rank"], cfg["qk_rope_head_dim"],
                                    cfg["qk_nope_head_dim"], cfg["v_head_dim"], cfg["intermediate_size"] // TP)
wb = 1 if args.quantization else 2
layer_w = ((qr + kr + rope) * h + H * (nope + rope) * qr + h * H * vd + 3 * inter * h) * wb + H * (nope + vd) * kr * 2
kv_b = 1 if args.kv_cache_dtype == "fp8_e4m3" else 2
layer_kv = bs * (ctx + warm + steps / 2) * (kr + rope) * kv_b
head = cfg["vocab_size"] // TP * h * 2
step_bytes = L * (layer_w + layer_kv) + head
rec = {"workload": f"DeepSeek-V3 dense part, one rank of TP={TP} (collectives stubbed), {L} layers, batch {bs}, context {ctx}, "
                   f"weights {'fp8 W8A8' if args.quantization else 'bf16'}, latent cache {'fp8' if kv_b == 1 else 'bf16'}, "
                   f"rope inside the decode launch: {args.fused_rope}",
       "ms_per_step": round(ms, 4), "us_per_layer_incl_lm_head_share": round(ms * 1e3 / L, 1),
       "algorithmic_bytes_per_step": int(step_bytes), "layer_weight_bytes": int(layer_w), "layer_kv_bytes": int(layer_kv),
       "hbm_gbs": round(step_bytes / ms / 1e6, 1), "frac_of_8TBs": round(step_bytes / ms / 1e6 / 8000.0, 4),
       "hip_graph": mr.graph_runner is not None}
print(json.dumps(rec), flush=True)
if args.out:
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
