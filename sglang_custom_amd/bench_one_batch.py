"""Static-batch latency harness with the semantics of the reference's
python/sglang/bench_one_batch.py (:211-265 synthetic inputs / extend / decode, :333-431
latency_test_run_once): prefill once, then decode step by step, synchronising around each."""
import time
from typing import List, Tuple

import numpy as np
import torch

from .managers.schedule_batch import Req, ScheduleBatch
from .model_executor.forward_batch_info import ForwardBatch
from .sampling.sampling_params import SamplingParams


def prepare_synthetic_inputs_for_latency_test(batch_size: int, input_len: int, seed: int = 0,
                                              vocab: int = 10000) -> List[Req]:
    rng = np.random.RandomState(seed)
    input_ids = rng.randint(0, vocab, (batch_size, input_len), dtype=np.int32)
    return [Req(rid=i, origin_input_ids=list(map(int, input_ids[i])),
                sampling_params=SamplingParams(temperature=0)) for i in range(batch_size)]


@torch.no_grad()
def extend(reqs, model_runner) -> Tuple[torch.Tensor, torch.Tensor, ScheduleBatch]:
    batch = ScheduleBatch.init_new(reqs=reqs, req_to_token_pool=model_runner.req_to_token_pool,
                                   token_to_kv_pool_allocator=model_runner.token_to_kv_pool_allocator,
                                   tree_cache=None, model_config=model_runner.model_config)
    batch.prepare_for_extend()
    forward_batch = ForwardBatch.init_new(batch.get_model_worker_batch(), model_runner)
    logits_output, _ = model_runner.forward(forward_batch)
    next_token_ids = model_runner.sample(logits_output, forward_batch)
    return next_token_ids, logits_output.next_token_logits, batch


@torch.no_grad()
def decode(input_token_ids, batch, model_runner) -> Tuple[torch.Tensor, torch.Tensor]:
    batch.output_ids = input_token_ids
    batch.prepare_for_decode()
    forward_batch = ForwardBatch.init_new(batch.get_model_worker_batch(), model_runner)
    logits_output, _ = model_runner.forward(forward_batch)
    next_token_ids = model_runner.sample(logits_output, forward_batch)
    return next_token_ids, logits_output.next_token_logits


def clear_pools(model_runner):
    model_runner.req_to_token_pool.clear()
    model_runner.token_to_kv_pool_allocator.clear()


def synchronize(device):
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()


def latency_test_run_once(model_runner, reqs, output_len: int):
    """Returns dict(prefill_latency, decode_latencies[], output_ids [bs, output_len])."""
    clear_pools(model_runner)
    dev = model_runner.device
    synchronize(dev)
    t0 = time.perf_counter()
    next_ids, _, batch = extend(reqs, model_runner)
    synchronize(dev)
    prefill_latency = time.perf_counter() - t0
    outs = [next_ids]
    decode_latencies = []
    for _ in range(output_len - 1):
        synchronize(dev)
        t0 = time.perf_counter()
        next_ids, _ = decode(next_ids, batch, model_runner)
        synchronize(dev)
        decode_latencies.append(time.perf_counter() - t0)
        outs.append(next_ids)
    return dict(prefill_latency=prefill_latency, decode_latencies=decode_latencies,
                output_ids=torch.stack([o.to(torch.int64) for o in outs], dim=1), batch=batch)
