"""Host-side logic on CPU (no GPU, no HIP library): Req/ScheduleBatch slot allocation, ForwardBatch,
ModelRunner, the Llama wiring, and tensor-parallel sharding over gloo with world_size 2 -- driven
through a CPU checker ops object and compared with the oracle model."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _build(cfg_name, dtype_name, tp_rank=0, tp_size=1, init_method=None):
    import cpu_ops
    from sglang_custom_amd import ops_backend
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.server_args import ServerArgs

    ops_backend.set_ops_backend(cpu_ops)
    cfg = hf_config_dict(cfg_name, dtype_name)
    dt = getattr(torch, dtype_name)
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=dt)
    sa = ServerArgs(dtype=dtype_name, device="cpu", max_total_tokens=512, max_running_requests=8,
                    disable_cuda_graph=True, dist_backend="gloo", dist_init_method=init_method)
    mr = ModelRunner(ModelConfig(cfg, dtype=dtype_name), sa, tp_rank=tp_rank, tp_size=tp_size, weights=w)
    return cfg, w, dt, mr


def _run(mr, prompts, steps):
    from sglang_custom_amd import bench_one_batch as bob
    from sglang_custom_amd.managers.schedule_batch import Req

    nxt, logits, batch = bob.extend([Req(i, p) for i, p in enumerate(prompts)], mr)
    toks, lg = [nxt.clone()], [logits.clone()]
    for _ in range(steps):
        nxt, logits = bob.decode(nxt, batch, mr)
        toks.append(nxt.clone())
        lg.append(logits.clone())
    return torch.stack(toks, 1), lg, batch


def _prompts(cfg):
    rng = np.random.RandomState(0)
    return [rng.randint(0, cfg["vocab_size"], n).tolist() for n in (5, 17, 32)]


def test_single_process_host_path_matches_oracle():
    from oracle.llama_ref import OracleLlama

    cfg, w, dt, mr = _build("tiny-llama-d64", "float32")
    prompts = _prompts(cfg)
    ref_toks, ref_logits = OracleLlama(cfg, w, dt, max_total_tokens=512).generate_greedy(prompts, 4)
    toks, lg, batch = _run(mr, prompts, 3)
    assert torch.equal(toks, ref_toks)
    for a, b in zip(lg, ref_logits):
        assert (a - b).abs().max() < 1e-4
    # slot bookkeeping (allocator.py:124-145): slot 0 is never handed out, prompts get contiguous
    # slots in request order, decode tokens are appended one slot per request per step
    r2t = mr.req_to_token_pool.req_to_token
    assert r2t[batch.req_pool_indices[0], :5].tolist() == [1, 2, 3, 4, 5]
    assert r2t[batch.req_pool_indices[1], :17].tolist() == list(range(6, 23))
    assert int(r2t[batch.req_pool_indices[0], 5]) == 55 and int(r2t[batch.req_pool_indices[1], 17]) == 56
    assert batch.seq_lens.tolist() == [5 + 3, 17 + 3, 32 + 3]


def _tp_worker(rank, world, port, q):
    try:
        torch.set_num_threads(2)
        cfg, w, dt, mr = _build("tiny-llama-d64", "float32", rank, world, f"tcp://127.0.0.1:{port}")
        toks, lg, _ = _run(mr, _prompts(cfg), 3)
        # sharding facts of this rank (linear.py:724-794, 1119-1303)
        layer = mr.model.model.layers[0]
        hq, hkv, d = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
        shapes = dict(qkv=tuple(layer.self_attn.qkv_proj.weight.shape), o=tuple(layer.self_attn.o_proj.weight.shape),
                      kv=tuple(mr.token_to_kv_pool.get_key_buffer(0).shape[1:]),
                      head=tuple(mr.model.lm_head.weight.shape))
        q.put((rank, toks.numpy(), torch.stack(lg).numpy(), shapes, (hq, hkv, d)))  # by value
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None))
        raise e


@pytest.mark.timeout(300)
def test_tensor_parallel_world2_gloo_matches_oracle():
    from oracle.llama_ref import OracleLlama
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == "error"), r[2]
    cfg = hf_config_dict("tiny-llama-d64", "float32")
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=torch.float32)
    ref_toks, ref_logits = OracleLlama(cfg, w, torch.float32, max_total_tokens=512).generate_greedy(_prompts(cfg), 4)
    for rank, toks, lg, shapes, (hq, hkv, d) in res:
        # every rank holds the full (all-gathered) logits and the same greedy tokens
        assert torch.equal(torch.from_numpy(toks), ref_toks), f"rank {rank}"
        assert (torch.from_numpy(lg) - torch.stack(ref_logits)).abs().max() < 1e-4
        hidden = cfg["hidden_size"]
        assert shapes["qkv"] == ((hq // 2 + 2 * max(1, hkv // 2)) * d, hidden)
        assert shapes["o"] == (hidden, hq // 2 * d)
        assert shapes["kv"] == (max(1, hkv // 2), d)
        assert shapes["head"][0] == cfg["vocab_size"] // 2


def _scheduler_case(mr, cfg, w, dt, tol):
    """Requests of different lengths arriving over time, a KV pool too small for all of them at
    once, prompts split into 16-token chunks: every request must produce exactly the tokens the
    oracle generates for it alone."""
    from oracle.llama_ref import OracleLlama
    from sglang_custom_amd.managers.schedule_batch import Req
    from sglang_custom_amd.managers.scheduler import Scheduler
    from sglang_custom_amd.sampling.sampling_params import SamplingParams

    rng = np.random.RandomState(3)
    spec = [(5, 6), (40, 3), (17, 9), (33, 1), (70, 4), (9, 12), (24, 5)]   # (prompt length, max_new_tokens)
    reqs = [Req(i, rng.randint(0, cfg["vocab_size"], n).tolist(), SamplingParams(max_new_tokens=m, temperature=0))
            for i, (n, m) in enumerate(spec)]
    sch = Scheduler(mr)
    for r in reqs[:3]:
        sch.add_request(r)
    arrivals = {2: reqs[3], 4: reqs[4], 5: reqs[5], 9: reqs[6]}
    it = 0
    while sch.waiting_queue or sch.chunked_req is not None or not sch.running_batch.is_empty() or arrivals:
        if it in arrivals:
            sch.add_request(arrivals.pop(it))
        progressed = sch.step()
        assert progressed or arrivals, "scheduler stalled"
        it += 1
        assert it < 500
    assert len(sch.finished) == len(reqs)
    assert sch.num_prefill_batches > len(reqs) // 2          # chunking + staggered arrivals: many prefills
    # every slot came back
    assert mr.token_to_kv_pool_allocator.available_size() == mr.max_total_num_tokens
    assert mr.req_to_token_pool.available_size() == mr.req_to_token_pool.size
    oracle = OracleLlama(cfg, w, dt, max_total_tokens=512)
    for r, (n, m) in zip(reqs, spec):
        ref_toks, _ = oracle.generate_greedy([r.origin_input_ids], m)
        assert r.output_ids == ref_toks[0].tolist(), (r.rid, r.output_ids, ref_toks[0].tolist())


def test_continuous_batching_scheduler_matches_isolated_generation():
    import cpu_ops
    from sglang_custom_amd import ops_backend
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.server_args import ServerArgs

    ops_backend.set_ops_backend(cpu_ops)
    cfg = hf_config_dict("tiny-llama-d64", "float32")
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=torch.float32)
    sa = ServerArgs(dtype="float32", device="cpu", max_total_tokens=160, max_running_requests=4,
                    disable_cuda_graph=True, chunked_prefill_size=16, max_prefill_tokens=64)
    mr = ModelRunner(ModelConfig(cfg, dtype="float32"), sa, weights=w)
    _scheduler_case(mr, cfg, w, torch.float32, 1e-4)
