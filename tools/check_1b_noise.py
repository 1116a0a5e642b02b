"""How far apart are three evaluations of the same Llama-3.2-1B-shaped prefill (128 tokens, synthetic
weights): the reference's bf16 CPU run (golden), this package's bf16 GPU run, and this package's fp32
GPU run on the same (bf16-valued) weights?  If both bf16 runs sit at a similar distance from the fp32
run, the bf16-vs-bf16 difference is rounding noise, not a defect.   python tools/check_1b_noise.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch

import cases
from sglang_custom_amd import bench_one_batch as bob
from sglang_custom_amd.configs.model_config import ModelConfig
from sglang_custom_amd.managers.schedule_batch import Req
from sglang_custom_amd.model_executor.model_runner import ModelRunner
from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
from sglang_custom_amd.server_args import ServerArgs

g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_llama-3.2-1b_bf16.npz"))
ref = torch.from_numpy(g["e2e/llama1b_bf16/prefill_logits"])
prompts = cases.e2e_prompts(10000, (128,))
out = {}
for tname in ("bfloat16", "float32"):
    cfg = hf_config_dict("llama-3.2-1b", tname)
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=torch.bfloat16)
    dt = getattr(torch, tname)
    sa = ServerArgs(dtype=tname, max_total_tokens=2048, max_running_requests=4, context_length=512, disable_cuda_graph=True)
    mr = ModelRunner(ModelConfig(cfg, dtype=tname, context_length=512), sa, weights={k: v.to(dt).cuda() for k, v in w.items()})
    _, logits, _ = bob.extend([Req(0, prompts[0])], mr)
    out[tname] = logits.float().cpu()
    del mr
    torch.cuda.empty_cache()
for a, b, n in ((ref, out["float32"], "reference bf16 (CPU) vs fp32"), (out["bfloat16"], out["float32"], "GPU bf16 vs fp32"),
                (ref, out["bfloat16"], "reference bf16 vs GPU bf16")):
    d = (a - b).abs()
    print(f"{n:34s} max {d.max():.4f}  mean {d.mean():.5f}  rms {d.pow(2).mean().sqrt():.5f}")
print("logit std", out["float32"].std().item(), "max", out["float32"].abs().max().item())
