"""Scheduler-driven throughput, plain loop vs overlap loop (the next step launched before the last
one's tokens are read back).  Not the contract benchmark: a small model on purpose, so that the
host work between steps is a visible share of a step.
usage: python tools/bench_scheduler.py [preset] [num_reqs] [prompt_len] [new_tokens]"""
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


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
    n_req = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    plen = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    new = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    cfg = hf_config_dict(preset, "bfloat16")
    sa = ServerArgs(dtype="bfloat16", max_total_tokens=n_req * (plen + new + 8) + 64, max_running_requests=n_req + 8,
                    context_length=plen + new + 16, cuda_graph_bs=[1, 2, 4, 8, 16, 32, 64][: max(1, n_req.bit_length())],
                    max_prefill_tokens=16384, schedule_conservativeness=1.5)
    mr = ModelRunner(ModelConfig(cfg, dtype="bfloat16", context_length=plen + new + 16), sa)
    rng = np.random.RandomState(0)
    prompts = [rng.randint(0, cfg["vocab_size"], plen).tolist() for _ in range(n_req)]
    for mode in ("plain", "overlap", "plain", "overlap"):
        sch = Scheduler(mr)
        for i, p in enumerate(prompts):
            sch.add_request(Req(i, p, SamplingParams(max_new_tokens=new, temperature=0, ignore_eos=True)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        (sch.run_until_done_overlap if mode == "overlap" else sch.run_until_done)()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        toks = sum(len(r.output_ids) for r in sch.finished.values())
        print(f"{preset} reqs {n_req} prompt {plen} new {new} {mode:8s}: {dt * 1e3:8.1f} ms  {toks / dt:9.0f} tok/s  "
              f"decode steps {sch.num_decode_steps}  ({dt * 1e3 / max(sch.num_decode_steps, 1):.3f} ms/step incl. prefill)", flush=True)


if __name__ == "__main__":
    main()
