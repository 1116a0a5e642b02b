"""Build container only (the reference tree must be importable): time the REFERENCE's own CPU decode
step -- its ModelRunner, LlamaForCausalLM, torch-native attention backend, sampler -- at the headline
configuration's batch and context (the reference's own number beside this build's CPU port).

    python tools/ref_cpu_decode_baseline.py [--model llama-3-8b] [--bs 64] [--ctx 1040] [--steps 5]

A 64 x 1024-token prefill on CPU would take a quarter of an hour, and decode time does not depend
on what the cache holds: the requests are prefilled with ONE token each, then their KV slots and
request-table rows are extended to `ctx` tokens (slots from the reference's own allocator, cache
contents random), and only decode steps are timed (wall clock per step, median; first step discarded).
"""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]

import _ref_harness  # noqa: E402

_ref_harness.install()
_ref_harness.use_amx_kernels(False)

import torch  # noqa: E402

from sglang_custom_amd.model_loader.synthetic import hf_config_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--ctx", type=int, default=1040)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()

    from sglang.srt.configs.model_config import ModelConfig
    from sglang.srt.managers.schedule_batch import Req, ScheduleBatch
    from sglang.srt.model_executor.forward_batch_info import ForwardBatch
    from sglang.srt.model_executor.model_runner import ModelRunner
    from sglang.srt.sampling.sampling_params import SamplingParams
    from sglang.srt.server_args import PortArgs, ServerArgs
    from sglang.srt.speculative.spec_info import SpeculativeAlgorithm

    d = tempfile.mkdtemp()
    cfg = hf_config_dict(args.model, "bfloat16")
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    bs, ctx = args.bs, args.ctx
    sa = ServerArgs(model_path=d, load_format="dummy", device="cpu", attention_backend="torch_native",
                    sampling_backend="pytorch", disable_cuda_graph=True, skip_tokenizer_init=True, dtype="bfloat16",
                    max_total_tokens=bs * (ctx + args.steps + 8), disable_radix_cache=True, chunked_prefill_size=-1,
                    mem_fraction_static=0.5, context_length=ctx + args.steps + 16, max_running_requests=bs + 4)
    pa = PortArgs.init_new(sa)
    mc = ModelConfig.from_server_args(sa)
    if getattr(mc.hf_config, "rope_theta", None) is None:
        mc.hf_config.rope_theta = cfg["rope_theta"]
    t0 = time.perf_counter()
    mr = ModelRunner(model_config=mc, mem_fraction_static=0.5, gpu_id=0, tp_rank=0, tp_size=1, pp_rank=0, pp_size=1,
                     nccl_port=pa.nccl_port, server_args=sa)
    print(f"model built in {time.perf_counter() - t0:.0f} s; threads {torch.get_num_threads()}", flush=True)

    reqs = []
    for i in range(bs):
        r = Req(rid=i, origin_input_text="", origin_input_ids=[1 + i],
                sampling_params=SamplingParams(temperature=0, max_new_tokens=args.steps + 2))
        r.prefix_indices = []
        r.fill_ids = r.origin_input_ids
        r.extend_input_len = 1
        r.logprob_start_len = 0
        reqs.append(r)
    batch = ScheduleBatch.init_new(reqs=reqs, req_to_token_pool=mr.req_to_token_pool,
                                   token_to_kv_pool_allocator=mr.token_to_kv_pool_allocator, tree_cache=None,
                                   model_config=mr.model_config, enable_overlap=False,
                                   spec_algorithm=SpeculativeAlgorithm.NONE, enable_custom_logit_processor=False)

    def step():
        fb = ForwardBatch.init_new(batch.get_model_worker_batch(), mr)
        lo, _ = mr.forward(fb)
        return mr.sample(lo, fb)

    with torch.no_grad():
        batch.prepare_for_extend()
        nxt = step()
        # grow every request to ctx - 1 cached tokens: slots from the reference's allocator, table rows written
        # the way prepare_for_extend writes them; cache contents random (timing does not depend on them)
        extra = ctx - 2
        slots = mr.token_to_kv_pool_allocator.alloc(bs * extra)
        assert slots is not None, "KV pool too small"
        table = mr.req_to_token_pool.req_to_token
        for i, r in enumerate(reqs):
            table[r.req_pool_idx, 1: 1 + extra] = slots[i * extra: (i + 1) * extra].to(table.dtype)
            r.fill_ids = r.fill_ids + [7] * extra
        batch.seq_lens = batch.seq_lens + extra
        batch.seq_lens_sum = int(batch.seq_lens.sum())
        for layer in range(mc.num_hidden_layers):
            mr.token_to_kv_pool.get_key_buffer(layer).normal_(0, 1)
            mr.token_to_kv_pool.get_value_buffer(layer).normal_(0, 1)
        times = []
        for s in range(args.steps + 1):
            batch.output_ids = nxt
            batch.prepare_for_decode()
            t0 = time.perf_counter()
            nxt = step()
            times.append(time.perf_counter() - t0)
            print(f"decode step {s}: context {int(batch.seq_lens[0])}, {times[-1] * 1e3:.0f} ms", flush=True)
    timed = times[1:]
    med = statistics.median(timed)
    print(json.dumps({"model": args.model, "bs": bs, "context": ctx, "backend": "torch_native", "dtype": "bfloat16",
                      "threads": torch.get_num_threads(), "steps_timed": len(timed),
                      "median_step_ms": round(med * 1e3, 1), "min_step_ms": round(min(timed) * 1e3, 1),
                      "max_step_ms": round(max(timed) * 1e3, 1), "tok_s": round(bs / med, 2)}))


if __name__ == "__main__":
    main()
