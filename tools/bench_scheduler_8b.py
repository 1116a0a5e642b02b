"""Continuous batching at the headline shape: Llama-3-8B bf16, 64 requests of 1024 prompt tokens and 128
new tokens through the Scheduler (admission, chunked/batched prefill, merge into the running decode batch,
HIP-graph decode, retirement), overlap loop on.  Two arrival patterns: all requests present at the start,
and 8 requests arriving every 6 loop iterations (the running batch grows while it decodes).  Reports
output tok/s over the whole run (prefill included), the decode-step time at full batch, and the host time
per loop iteration that is NOT spent waiting for the device.  Writes one JSON object per pattern.
usage: [KV_LAYOUT=auto|token_major|head_major] [MAX_PREFILL_TOKENS=16384] [GRAPH_BS=8,16,...,64] python tools/bench_scheduler_8b.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sglang_custom_amd.configs.model_config import ModelConfig
from sglang_custom_amd.managers.schedule_batch import Req
from sglang_custom_amd.managers.scheduler import Scheduler
from sglang_custom_amd.model_executor.model_runner import ModelRunner
from sglang_custom_amd.model_loader.synthetic import hf_config_dict
from sglang_custom_amd.sampling.sampling_params import SamplingParams
from sglang_custom_amd.server_args import ServerArgs

N_REQ, PLEN, NEW = 64, 1024, 128


class TimedScheduler(Scheduler):
    """Adds two clocks: time spent waiting for the device inside _collect, and per-iteration records."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.wait_s = 0.0
        self.full_batch_steps = []   # (t_collect_done) of decode steps with all requests running

    def _collect(self, launched):
        if launched["done"] is not None:
            t0 = time.perf_counter()
            launched["done"].synchronize()
            self.wait_s += time.perf_counter() - t0
        if not launched["extend"] and len(launched["reqs"]) == N_REQ:
            self.full_batch_steps.append(time.perf_counter())
        super()._collect(launched)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    cfg = hf_config_dict("llama-3-8b", "bfloat16")
    sa = ServerArgs(dtype="bfloat16", max_total_tokens=N_REQ * (PLEN + NEW + 8) + 64, max_running_requests=N_REQ + 8,
                    context_length=PLEN + NEW + 16, cuda_graph_bs=[int(x) for x in os.environ.get("GRAPH_BS", "8,16,24,32,40,48,56,64").split(",")],
                    max_prefill_tokens=int(os.environ.get("MAX_PREFILL_TOKENS", "16384")), schedule_conservativeness=1.5, kv_layout=os.environ.get("KV_LAYOUT", "auto"))
    mr = ModelRunner(ModelConfig(cfg, dtype="bfloat16", context_length=PLEN + NEW + 16), sa)
    rng = np.random.RandomState(0)
    prompts = [rng.randint(0, cfg["vocab_size"], PLEN).tolist() for _ in range(N_REQ)]

    def reqs(lo, hi):
        return [Req(i, prompts[i], SamplingParams(max_new_tokens=NEW, temperature=0, ignore_eos=True)) for i in range(lo, hi)]

    results = []
    for pattern in ("warm-up", "all at start", "8 requests every 6 iterations"):
        sch = TimedScheduler(mr)
        if pattern == "8 requests every 6 iterations":
            arrivals = {6 * j: reqs(8 * j, 8 * j + 8) for j in range(N_REQ // 8)}
        else:
            arrivals = {0: reqs(0, N_REQ)}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sch.run_until_done_overlap(arrivals=arrivals)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        toks = sum(len(r.output_ids) for r in sch.finished.values())
        assert len(sch.finished) == N_REQ and toks == N_REQ * NEW, (len(sch.finished), toks)
        iters = sch.num_decode_steps + sch.num_prefill_batches
        fb = sch.full_batch_steps
        full_ms = (fb[-1] - fb[0]) / (len(fb) - 1) * 1e3 if len(fb) > 2 else None
        rec = {"model": "llama-3-8b bf16", "requests": N_REQ, "prompt_len": PLEN, "new_tokens": NEW, "arrivals": pattern,
               "loop": "overlap (next step launched before the last step's ids are read)", "hip_graph": True,
               "wall_ms": round(dt * 1e3, 1), "output_tok_s_incl_prefill": round(toks / dt, 1),
               "prefill_batches": sch.num_prefill_batches, "decode_steps": sch.num_decode_steps,
               "decode_ms_per_step_at_batch_64": None if full_ms is None else round(full_ms, 3),
               "decode_tok_s_at_batch_64": None if full_ms is None else round(N_REQ / full_ms * 1e3, 1),
               "host_ms_per_iteration_not_waiting_for_device": round((dt - sch.wait_s) / max(iters, 1) * 1e3, 3),
               "device_wait_share": round(sch.wait_s / dt, 3)}
        print(json.dumps(rec), flush=True)
        if pattern != "warm-up":
            results.append(rec)
    if out_path:
        with open(out_path, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
