"""Host-side profile (cProfile) of the overlap scheduler loop on a small model: where the Python
time per decode step goes.  usage: python tools/profile_scheduler_host.py"""
import cProfile
import os
import pstats
import sys

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

cfg = hf_config_dict("llama-3.2-1b", "bfloat16")
sa = ServerArgs(dtype="bfloat16", max_total_tokens=16 * 300, max_running_requests=24, context_length=300,
                cuda_graph_bs=[16], max_prefill_tokens=16384, schedule_conservativeness=1.5)
mr = ModelRunner(ModelConfig(cfg, dtype="bfloat16", context_length=300), sa)
rng = np.random.RandomState(0)
prompts = [rng.randint(0, cfg["vocab_size"], 64).tolist() for _ in range(16)]


def run():
    sch = Scheduler(mr)
    for i, p in enumerate(prompts):
        sch.add_request(Req(i, p, SamplingParams(max_new_tokens=200, temperature=0, ignore_eos=True)))
    sch.run_until_done_overlap()
    torch.cuda.synchronize()


run()
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
