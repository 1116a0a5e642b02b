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


def _build(cfg_name, dtype_name, tp_rank=0, tp_size=1, init_method=None, kv_layout="auto"):
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
                    disable_cuda_graph=True, dist_backend="gloo", dist_init_method=init_method, kv_layout=kv_layout)
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


def test_head_major_kv_pool_is_the_same_tensor_to_every_reader():
    """kv_layout="head_major": the memory behind a layer is [heads, slots, 2, d] (keys | values), what get_key_buffer /
    get_value_buffer hand out is still [slots, heads, d]; the host path over it gives the oracle's tokens and logits,
    and the rows land where the views say.  "auto" keeps the reference's layout on the CPU; other values are refused."""
    from oracle.llama_ref import OracleLlama

    cfg, w, dt, mr = _build("tiny-llama-d64", "float32", kv_layout="head_major")
    pool = mr.token_to_kv_pool
    kb, vb = pool.get_kv_buffer(0)
    heads, d = kb.shape[1], kb.shape[2]
    assert pool.head_major and kb.shape == vb.shape and kb.shape[0] >= pool.size + pool.page_size   # (a few unused rows may pad the head stride)
    assert kb.stride() == (2 * d, 2 * kb.shape[0] * d, 1) and vb.data_ptr() == kb.data_ptr() + d * kb.element_size()
    prompts = _prompts(cfg)
    ref_toks, ref_logits = OracleLlama(cfg, w, dt, max_total_tokens=512).generate_greedy(prompts, 4)
    toks, lg, batch = _run(mr, prompts, 3)
    assert torch.equal(toks, ref_toks)
    for a, b in zip(lg, ref_logits):
        assert (a - b).abs().max() < 1e-4
    # the same run over the reference's layout leaves the same rows behind the same view
    cfg2, w2, dt2, mr2 = _build("tiny-llama-d64", "float32", kv_layout="token_major")
    assert not mr2.token_to_kv_pool.head_major and mr2.token_to_kv_pool.get_key_buffer(0).is_contiguous()
    _run(mr2, prompts, 3)
    for layer in range(cfg["num_hidden_layers"]):
        k1, v1 = pool.get_kv_buffer(layer)
        k2, v2 = mr2.token_to_kv_pool.get_kv_buffer(layer)
        n = k2.shape[0]
        assert torch.equal(k1[1:n], k2[1:]) and torch.equal(v1[1:n], v2[1:])
    assert not _build("tiny-llama-d64", "float32")[3].token_to_kv_pool.head_major     # "auto" on the CPU
    with pytest.raises(ValueError):
        _build("tiny-llama-d64", "float32", kv_layout="sideways")


def _tp_worker(rank, world, port, q):
    try:
        torch.set_num_threads(2 if world <= 2 else 1)
        cfg, w, dt, mr = _build("tiny-llama-d64", "float32", rank, world, f"tcp://127.0.0.1:{port}")
        toks, lg, _ = _run(mr, _prompts(cfg), 3)
        # sharding facts of this rank (linear.py:724-794, 1119-1303)
        layer = mr.model.model.layers[0]
        hq, hkv, d = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
        shapes = dict(qkv=tuple(layer.self_attn.qkv_proj.weight.shape), o=tuple(layer.self_attn.o_proj.weight.shape),
                      kv=tuple(mr.token_to_kv_pool.get_key_buffer(0).shape[1:]),
                      head=tuple(mr.model.lm_head.weight.shape))
        # host-side agreement helpers and the token-id sync (sampler.py:137-149 of the reference)
        from sglang_custom_amd.distributed import parallel_state as ps

        ps.tp_barrier()
        assert ps.tp_any(rank == 1) is True and ps.tp_any(False) is False
        ids = torch.tensor([5, 3, 9] if rank == 0 else [4, 7, 9], dtype=torch.int64)
        assert ps.sync_token_ids_across_tp(ids).tolist() == [4, 3, 9]
        q.put((rank, toks.numpy(), torch.stack(lg).numpy(), shapes, (hq, hkv, d)))  # by value
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None))
        raise e


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_tensor_parallel_gloo_matches_oracle(world):
    """8 query heads over 2 kv heads: at 4 and 8 ranks a kv head is replicated (rank r owns kv head r // replicas,
    layers/linear.py:778-783 of the reference) and at 8 ranks every rank holds one query head."""
    from oracle.llama_ref import OracleLlama
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker, args=(r, world, port, q)) for r in range(world)]
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
        assert shapes["qkv"] == ((hq // world + 2 * max(1, hkv // world)) * d, hidden)
        assert shapes["o"] == (hidden, hq // world * d)
        assert shapes["kv"] == (max(1, hkv // world), d)
        assert shapes["head"][0] == cfg["vocab_size"] // world


def _tp_worker_deepseek(rank, world, port, preset, q):
    try:
        torch.set_num_threads(2)
        import cases
        from sglang_custom_amd import bench_one_batch as bob
        from sglang_custom_amd.managers.schedule_batch import Req

        cfg, w, dt, mr = _build(preset, "float32", rank, world, f"tcp://127.0.0.1:{port}")
        prompts = cases.e2e_prompts(cfg["vocab_size"], (5, 33))
        nxt, lg, batch = bob.extend([Req(i, p) for i, p in enumerate(prompts)], mr)
        toks, logits = [nxt.clone()], [lg.clone()]
        for _ in range(3):
            nxt, lg = bob.decode(nxt, batch, mr)
            toks.append(nxt.clone())
            logits.append(lg.clone())
        at = mr.model.model.layers[0].self_attn
        shapes = dict(q=tuple((at.q_b_proj if at.q_lora_rank is not None else at.q_proj).weight.shape),
                      kv_b=tuple(at.kv_b_proj.weight.shape), o=tuple(at.o_proj.weight.shape),
                      w_kc=tuple(at.w_kc.shape), w_vc=tuple(at.w_vc.shape),
                      gate_up=tuple(mr.model.model.layers[0].mlp.gate_up_proj.weight.shape),
                      cache=tuple(mr.token_to_kv_pool.get_key_buffer(0).shape[1:]),
                      heads=(at.attn_mqa.tp_q_head_num, at.attn_mha.tp_k_head_num))
        q.put((rank, torch.stack(toks, 1).numpy(), torch.stack(logits).numpy(), shapes))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc(), None))
        raise e


@pytest.mark.timeout(300)
@pytest.mark.parametrize("preset", ["tiny-deepseek", "tiny-deepseek-yarn"])
def test_deepseek_tensor_parallel_world2_gloo_matches_oracle(preset):
    """The MLA stack over two ranks as the reference shards it (models/deepseek_v2.py:781-840, 182-198): heads
    divided (q_b_proj / q_proj / kv_b_proj column-parallel, o_proj row-parallel), the low-rank projections and
    the latent cache replicated, the MLP column- / row-parallel.  Both ranks must produce the single-process
    oracle's tokens and logits (prefill in the multi-head form, decode in the absorbed form)."""
    import cases
    from oracle.deepseek_ref import OracleDeepseek
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker_deepseek, args=(r, 2, port, preset, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == "error"), r[2]
    cfg = hf_config_dict(preset, "float32")
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=torch.float32)
    prompts = cases.e2e_prompts(cfg["vocab_size"], (5, 33))
    ref_toks, ref_logits = OracleDeepseek(cfg, w, torch.float32, max_total_tokens=512, max_reqs=8,
                                          context_len=256).generate_greedy(prompts, 4)
    H, nope, rope, vd, rank_ = (cfg["num_attention_heads"], cfg["qk_nope_head_dim"], cfg["qk_rope_head_dim"],
                                cfg["v_head_dim"], cfg["kv_lora_rank"])
    for rank, toks, lg, shapes in res:
        assert torch.equal(torch.from_numpy(toks), ref_toks), f"rank {rank}"
        assert (torch.from_numpy(lg) - torch.stack(ref_logits)).abs().max() < 1e-4
        q_in = cfg["q_lora_rank"] or cfg["hidden_size"]
        assert shapes["q"] == (H // 2 * (nope + rope), q_in)
        assert shapes["kv_b"] == (H // 2 * (nope + vd), rank_)
        assert shapes["o"] == (cfg["hidden_size"], H // 2 * vd)
        assert shapes["w_kc"] == (H // 2, nope, rank_) and shapes["w_vc"] == (H // 2, rank_, vd)
        assert shapes["gate_up"] == (cfg["intermediate_size"], cfg["hidden_size"])      # 2 * inter / 2 ranks
        assert shapes["cache"] == (1, rank_ + rope)                                      # replicated latent rows
        assert shapes["heads"] == (H // 2, H // 2)


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
    return sch


@pytest.mark.parametrize("mixed", [False, True])
def test_continuous_batching_scheduler_matches_isolated_generation(mixed):
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
                    disable_cuda_graph=True, chunked_prefill_size=16, max_prefill_tokens=64,
                    enable_mixed_chunk=mixed)
    mr = ModelRunner(ModelConfig(cfg, dtype="float32"), sa, weights=w)
    sch = _scheduler_case(mr, cfg, w, torch.float32, 1e-4)
    # with mixing on, decode requests rode along prefill chunks instead of waiting for them
    assert (sch.num_mixed_batches > 0) == mixed
    if mixed:
        assert sch.num_decode_steps < 30


@pytest.mark.parametrize("radix,page", [(False, 1), (True, 1), (True, 4), (False, 4)])
def test_scheduler_retracts_decoding_requests_when_the_pool_runs_out(radix, page):
    """Admission reserves only a fraction of what running requests may still generate, so a pool
    that cannot hold every admitted request to its end runs short mid-decode: requests are
    retracted (slots back, outputs kept), re-prefilled over prompt + outputs later, and still
    produce exactly the tokens the oracle generates for them alone."""
    import cpu_ops
    from oracle.llama_ref import OracleLlama
    from sglang_custom_amd import ops_backend
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.managers.schedule_batch import Req
    from sglang_custom_amd.managers.scheduler import Scheduler
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.sampling.sampling_params import SamplingParams
    from sglang_custom_amd.server_args import ServerArgs

    ops_backend.set_ops_backend(cpu_ops)
    cfg = hf_config_dict("tiny-llama-d64", "float32")
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=torch.float32)
    # pages: requests are admitted against whole pages plus one, so over-committing the pool takes
    # longer outputs and a bolder estimate
    sa = ServerArgs(dtype="float32", device="cpu", max_total_tokens=160 if page == 1 else 256, max_running_requests=6,
                    disable_cuda_graph=True, max_prefill_tokens=128, page_size=page,
                    schedule_conservativeness=1.0 if page == 1 else 0.1)
    mr = ModelRunner(ModelConfig(cfg, dtype="float32"), sa, weights=w)
    rng = np.random.RandomState(11)
    spec = [(20, 40), (24, 40), (18, 44), (22, 36), (10, 30)]
    if page > 1:
        spec = [(n, 2 * m) for n, m in spec]
    shared = rng.randint(0, cfg["vocab_size"], 8).tolist()
    reqs = [Req(i, shared + rng.randint(0, cfg["vocab_size"], n - 8).tolist(),
                SamplingParams(max_new_tokens=m, temperature=0)) for i, (n, m) in enumerate(spec)]
    sch = Scheduler(mr, enable_radix_cache=radix)
    assert sch.new_token_ratio == pytest.approx(0.7 if page == 1 else 0.07)
    for r in reqs:
        sch.add_request(r)
    peak_running, ratios = 0, set()
    it = 0
    while sch.waiting_queue or not sch.running_batch.is_empty():
        assert sch.step(), "scheduler stalled"
        peak_running = max(peak_running, len(sch.running_batch.reqs))
        ratios.add(round(sch.new_token_ratio, 6))
        it += 1
        assert it < 2000
    assert len(sch.finished) == len(reqs)
    assert sch.total_retracted_reqs > 0 and any(r.is_retracted for r in reqs)
    assert peak_running >= 3                      # more were admitted than fit to the end (3 x ~62 > 160)
    assert len(ratios) > 2                        # the estimate decayed and was re-estimated
    alloc = mr.token_to_kv_pool_allocator
    assert alloc.available_size() + sch.tree_cache.evictable_size() == mr.max_total_num_tokens
    assert sch.tree_cache.protected_size() == 0
    assert mr.req_to_token_pool.available_size() == mr.req_to_token_pool.size
    oracle = OracleLlama(cfg, w, torch.float32, max_total_tokens=512)
    for r, (n, m) in zip(reqs, spec):
        ref_toks, _ = oracle.generate_greedy([r.origin_input_ids], m)
        assert r.output_ids == ref_toks[0].tolist(), (r.rid, r.output_ids, ref_toks[0].tolist())


@pytest.mark.parametrize("radix,page,tight", [(False, 1, False), (True, 1, False), (True, 4, False), (False, 1, True)])
def test_overlap_scheduling_matches_isolated_generation(radix, page, tight):
    """The loop that launches step N+1 before reading step N's tokens: requests finish one step
    late (their extra step's slot is returned too), chunked prompts, staggered arrivals, the prefix
    tree with and without pages, and (tight) a pool small enough to force retraction while results
    are in flight -- every request still gets exactly the oracle's tokens for its prompt alone."""
    import cpu_ops
    from oracle.llama_ref import OracleLlama
    from sglang_custom_amd import ops_backend
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.managers.schedule_batch import Req
    from sglang_custom_amd.managers.scheduler import Scheduler
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.sampling.sampling_params import SamplingParams
    from sglang_custom_amd.server_args import ServerArgs

    ops_backend.set_ops_backend(cpu_ops)
    cfg = hf_config_dict("tiny-llama-d64", "float32")
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=torch.float32)
    sa = ServerArgs(dtype="float32", device="cpu", max_total_tokens=176 if tight else 320, max_running_requests=6,
                    disable_cuda_graph=True, chunked_prefill_size=None if tight else 24, max_prefill_tokens=128,
                    page_size=page)
    mr = ModelRunner(ModelConfig(cfg, dtype="float32"), sa, weights=w)
    rng = np.random.RandomState(5)
    spec = ([(20, 40), (24, 40), (18, 44), (22, 36), (10, 30)] if tight else
            [(5, 6), (41, 3), (17, 9), (33, 1), (70, 4), (9, 12), (24, 5), (30, 2)])
    stem = rng.randint(0, cfg["vocab_size"], 2 * page + 3).tolist()
    reqs = []
    for i, (n, m) in enumerate(spec):
        ids = rng.randint(0, cfg["vocab_size"], n).tolist()
        if radix and n > len(stem):
            ids[: len(stem)] = stem
        reqs.append(Req(i, ids, SamplingParams(max_new_tokens=m, temperature=0)))
    sch = Scheduler(mr, enable_radix_cache=radix)
    for r in (reqs if tight else reqs[:3]):
        sch.add_request(r)
    arrivals = {} if tight else {2: [reqs[3]], 4: [reqs[4]], 5: [reqs[5]], 9: reqs[6:]}
    sch.run_until_done_overlap(max_iters=3000, arrivals=arrivals)
    assert len(sch.finished) == len(reqs)
    if tight:
        assert sch.total_retracted_reqs > 0
    alloc = mr.token_to_kv_pool_allocator
    assert alloc.available_size() + sch.tree_cache.evictable_size() == mr.max_total_num_tokens
    assert sch.tree_cache.protected_size() == 0
    assert mr.req_to_token_pool.available_size() == mr.req_to_token_pool.size
    oracle = OracleLlama(cfg, w, torch.float32, max_total_tokens=512)
    for r, (n, m) in zip(reqs, spec):
        ref_toks, _ = oracle.generate_greedy([r.origin_input_ids], m)
        assert r.output_ids == ref_toks[0].tolist(), (r.rid, r.output_ids, ref_toks[0].tolist())


def test_radix_cache_tree_semantics():
    """match / insert / split / lock / LRU eviction of the prefix tree, on a fake allocator."""
    from sglang_custom_amd.mem_cache.radix_cache import RadixCache

    class Alloc:
        device = "cpu"

        def __init__(self):
            self.freed = []

        def free(self, idx):
            self.freed.extend(idx.tolist())

    alloc = Alloc()
    tree = RadixCache(req_to_token_pool=None, token_to_kv_pool_allocator=alloc)
    a = [1, 2, 3, 4, 5, 6]
    assert tree.insert(a, torch.arange(10, 16)) == 0
    v, node = tree.match_prefix([1, 2, 3, 9])
    assert v.tolist() == [10, 11, 12] and node.key == [1, 2, 3]          # node split at the match
    assert tree.insert([1, 2, 3, 7, 8], torch.tensor([90, 91, 92, 17, 18])) == 3   # 3 duplicates reported
    assert tree.total_size() == 8 and tree.evictable_size() == 8
    v, leaf = tree.match_prefix([1, 2, 3, 7, 8])
    assert v.tolist() == [10, 11, 12, 17, 18]
    tree.inc_lock_ref(leaf)
    assert tree.evictable_size() == 3 and tree.protected_size() == 5
    assert tree.evict(100) == 3 and sorted(alloc.freed) == [13, 14, 15]   # only the unlocked branch goes
    tree.dec_lock_ref(leaf)
    assert tree.evictable_size() == 5
    assert tree.evict(1) == 2 and sorted(alloc.freed)[-2:] == [17, 18]    # leaf first, parent stays
    assert tree.evict(100) == 3 and tree.total_size() == 0
    assert tree.match_prefix([1, 2])[0].numel() == 0


def test_scheduler_with_radix_prefix_cache_matches_isolated_generation():
    """Requests that share prompt prefixes, arriving after earlier ones finished or mid-run: the
    cached slots are reused (extend with prefix), outputs equal the oracle's for each prompt
    alone, every slot is either free or owned by the tree at the end, and a pool that is too
    small for everything forces LRU eviction."""
    import cpu_ops
    from oracle.llama_ref import OracleLlama
    from sglang_custom_amd import ops_backend
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.managers.schedule_batch import Req
    from sglang_custom_amd.managers.scheduler import Scheduler
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.sampling.sampling_params import SamplingParams
    from sglang_custom_amd.server_args import ServerArgs

    ops_backend.set_ops_backend(cpu_ops)
    cfg = hf_config_dict("tiny-llama-d64", "float32")
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=torch.float32)
    sa = ServerArgs(dtype="float32", device="cpu", max_total_tokens=120, max_running_requests=4,
                    disable_cuda_graph=True, chunked_prefill_size=32, max_prefill_tokens=64)
    mr = ModelRunner(ModelConfig(cfg, dtype="float32"), sa, weights=w)
    rng = np.random.RandomState(5)
    sys_prompt = rng.randint(0, cfg["vocab_size"], 40).tolist()
    other = rng.randint(0, cfg["vocab_size"], 45).tolist()
    prompts = [sys_prompt + rng.randint(0, cfg["vocab_size"], 7).tolist(),
               sys_prompt + rng.randint(0, cfg["vocab_size"], 12).tolist(),
               other,
               sys_prompt[:25] + rng.randint(0, cfg["vocab_size"], 9).tolist(),
               sys_prompt + rng.randint(0, cfg["vocab_size"], 3).tolist(),
               other + rng.randint(0, cfg["vocab_size"], 5).tolist()]
    new_tokens = [5, 4, 6, 3, 8, 4]
    reqs = [Req(i, p, SamplingParams(max_new_tokens=m, temperature=0)) for i, (p, m) in enumerate(zip(prompts, new_tokens))]
    sch = Scheduler(mr, enable_radix_cache=True)
    sch.add_request(reqs[0])
    arrivals = {3: reqs[1], 4: reqs[2], 9: reqs[3], 14: reqs[4], 22: reqs[5]}
    it = 0
    while sch.waiting_queue or sch.chunked_req is not None or not sch.running_batch.is_empty() or arrivals:
        if it in arrivals:
            sch.add_request(arrivals.pop(it))
        assert sch.step() or arrivals
        it += 1
        assert it < 500
    assert len(sch.finished) == len(reqs)
    assert sch.cached_prefix_tokens >= 40 + 25          # at least the shared system prompt was reused
    assert sch.evicted_tokens > 0                       # 120 slots do not hold every prompt: LRU eviction ran
    tree, alloc = sch.tree_cache, mr.token_to_kv_pool_allocator
    assert alloc.available_size() + tree.total_size() == mr.max_total_num_tokens
    assert tree.protected_size() == 0 and tree.evictable_size() == tree.total_size()
    assert mr.req_to_token_pool.available_size() == mr.req_to_token_pool.size
    oracle = OracleLlama(cfg, w, torch.float32, max_total_tokens=512)
    for r, m in zip(reqs, new_tokens):
        ref_toks, _ = oracle.generate_greedy([r.origin_input_ids], m)
        assert r.output_ids == ref_toks[0].tolist(), (r.rid, r.output_ids, ref_toks[0].tolist())
    # everything the tree holds can be evicted back to the allocator
    held = tree.total_size()
    assert tree.evict(10 ** 9) == held
    assert alloc.available_size() == mr.max_total_num_tokens and tree.total_size() == 0


@pytest.mark.parametrize("q_lora_rank", [None, 48])
def test_mla_oracle_absorbed_equals_expanded(q_lora_rank):
    """The two restatements of multi-head latent attention (latent expanded per head vs. kv_b_proj
    folded into query and output) are one function: equal to float64 rounding."""
    from oracle import mla_ref

    g = torch.Generator().manual_seed(2)
    hidden, H, nope, rope, vd, kr = 64, 4, 16, 8, 12, 32
    w = {"kv_a_layernorm": 1 + 0.1 * torch.randn(kr, generator=g), "kv_b_proj": torch.randn(H * (nope + vd), kr, generator=g) * 0.2,
         "o_proj": torch.randn(hidden, H * vd, generator=g) * 0.2}
    if q_lora_rank is None:
        w["q_proj"] = torch.randn(H * (nope + rope), hidden, generator=g) * 0.2
        w["kv_a_proj_with_mqa"] = torch.randn(kr + rope, hidden, generator=g) * 0.2
    else:
        w["fused_qkv_a_proj_with_mqa"] = torch.randn(q_lora_rank + kr + rope, hidden, generator=g) * 0.2
        w["q_a_layernorm"] = 1 + 0.1 * torch.randn(q_lora_rank, generator=g)
        w["q_b_proj"] = torch.randn(H * (nope + rope), q_lora_rank, generator=g) * 0.2
    x = torch.randn(11, hidden, generator=g)
    a = mla_ref.mla_attention(x, w, H, nope, rope, vd, q_lora_rank, kr)
    b = mla_ref.mla_attention_absorbed(x, w, H, nope, rope, vd, q_lora_rank, kr)
    assert a.shape == (11, hidden) and (a - b).abs().max().item() < 1e-10 * a.abs().max().item()


def test_mla_oracle_rotation_is_the_pinned_interleaved_rope():
    """oracle/mla_ref.py rotates with its own float64 helper; it must be the rotation of
    ops_ref.apply_rope(is_neox=False), which the reference's golden vectors pin."""
    from oracle import mla_ref, ops_ref

    g = torch.Generator().manual_seed(4)
    T, H, R, theta = 9, 3, 64, 10000.0
    x = torch.randn(T, H, R, generator=g, dtype=torch.float64)
    pos = torch.tensor([0, 1, 2, 5, 17, 100, 511, 3, 64])
    cache = ops_ref.cos_sin_cache(R, 512, theta, None, torch.float64)
    ang = pos.double()[:, None] * (1.0 / theta ** (torch.arange(0, R, 2, dtype=torch.float64) / R))[None]
    cache64 = torch.cat([ang.cos(), ang.sin()], -1)                      # exact angles for the rows in use
    assert (cache[pos] - cache64).abs().max().item() < 1e-4              # same table up to its float32 construction
    table = torch.zeros(512, R, dtype=torch.float64)
    table[pos] = cache64
    q, _ = ops_ref.apply_rope(pos, x.reshape(T, H * R), x[:, :1].reshape(T, R), R, table, False)
    assert (q.view(T, H, R) - mla_ref._rope_gptj(x, pos, theta)).abs().max().item() < 1e-12


def test_bench_spawns_one_live_rank_per_gpu_and_refuses_to_mislabel():
    """`python bench.py --gpus N` without a launcher starts N fresh rank processes (rendezvous
    rehearsed over gloo, no GPU); a rank count that differs from --gpus is an error, never a line
    that claims N GPUs."""
    import json
    import subprocess
    import sys

    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_alive"] == 2 and line["scaling"] == "strong"
    assert line["config"]["parallelism"] == "tp2" and "TP=2" in line["metric"]
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing to mislabel" in r.stderr and "{" not in r.stdout


class _Only:
    """A stand-in that exposes exactly the attributes the reference's object of that kind has."""

    def __init__(self, obj, allowed):
        object.__setattr__(self, "_obj", obj)
        object.__setattr__(self, "_allowed", set(allowed))

    def __getattr__(self, name):
        if name not in self._allowed:
            raise AttributeError(f"the reference's object has no attribute {name!r}")
        return getattr(self._obj, name)

    def __setattr__(self, name, value):
        setattr(self._obj, name, value)


def test_backend_is_a_drop_in_for_reference_shaped_objects():
    """tests/golden/dropin.json records a run made where the reference is importable: the reference's
    ModelRunner / RadixAttention / ForwardBatch / MHATokenToKVPool drove Mi355AttnBackend (assigned to
    model_runner.attn_backend) and produced the reference backend's tokens and logits.  Here the same
    backend runs on stand-ins that expose exactly the attribute surface recorded there, so a
    dependency on anything the reference's objects lack fails with AttributeError."""
    import json
    from types import SimpleNamespace

    import cases
    import cpu_ops
    from oracle import ops_ref
    from sglang_custom_amd import ops_backend
    from sglang_custom_amd.layers.attention.mi355_backend import Mi355AttnBackend
    from sglang_custom_amd.layers.radix_attention import RadixAttention
    from sglang_custom_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool
    from sglang_custom_amd.model_executor.forward_batch_info import ForwardBatch, ForwardMode

    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dropin.json")))
    assert rec["tokens_equal_reference_backend"] and rec["max_abs_logit_difference"] < 1e-5
    assert rec["mi355_backend_signatures"] == rec["reference_interface_signatures"]
    import inspect

    for m, params in rec["reference_interface_signatures"].items():      # and the code still has those signatures
        assert list(inspect.signature(getattr(Mi355AttnBackend, m)).parameters) == params, m
    attrs = rec["reference_attrs"]
    ops_backend.set_ops_backend(cpu_ops)
    try:
        dt, Hq, Hkv, D = torch.float32, 8, 2, 64
        pre, ext = [6, 0, 11], [9, 14, 1]
        seq = [a + b for a, b in zip(pre, ext)]
        kc, vc, r2t, rows = cases.paged_inputs(400, dt, 3, Hkv, D, [n + 1 for n in seq])
        pool = MHATokenToKVPool(kc.shape[0] - 1, 1, dt, Hkv, D, 1, "cpu")
        pool.get_key_buffer(0).copy_(kc)
        pool.get_value_buffer(0).copy_(vc)
        r2tp = ReqToTokenPool(r2t.shape[0], r2t.shape[1], "cpu")
        r2tp.req_to_token.copy_(r2t)
        spool, sr2t = _Only(pool, attrs["token_to_kv_pool"]), _Only(r2tp, attrs["req_to_token_pool"])
        cfg = _Only(SimpleNamespace(num_attention_heads=Hq, num_key_value_heads=Hkv, context_len=64, head_dim=D, vocab_size=32),
                    attrs["model_config"])
        runner = _Only(SimpleNamespace(device="cpu", tp_size=1, req_to_token_pool=sr2t, token_to_kv_pool=spool, model_config=cfg,
                                       sliding_window_size=None, attn_backend=None), attrs["model_runner"])
        be = Mi355AttnBackend(runner)
        layer = _Only(RadixAttention(Hq, D, D ** -0.5, Hkv, 0), attrs["radix_attention"])
        q = cases.randn((sum(ext), Hq * D), 401, dt)
        loc = torch.cat([r2t[rows[b], pre[b]: seq[b]].long() for b in range(3)])
        knew, vnew = kc[loc].clone(), vc[loc].clone()

        def mode(m):
            return _Only(m, attrs["forward_mode"])

        fb = _Only(ForwardBatch(forward_mode=mode(ForwardMode.EXTEND), batch_size=3, input_ids=torch.zeros(sum(ext), dtype=torch.int64),
                                req_pool_indices=rows, seq_lens=torch.tensor(seq), out_cache_loc=loc, seq_lens_sum=sum(seq),
                                seq_lens_cpu=torch.tensor(seq), extend_seq_lens=torch.tensor(ext, dtype=torch.int32),
                                extend_prefix_lens=torch.tensor(pre, dtype=torch.int32), extend_prefix_lens_cpu=pre,
                                extend_seq_lens_cpu=ext, req_to_token_pool=sr2t, token_to_kv_pool=spool, attn_backend=be),
                   attrs["forward_batch"])
        be.init_forward_metadata(fb)
        o = be.forward(q, knew, vnew, layer, fb)
        want = ops_ref.extend_attention(q.view(-1, Hq, D), kc, vc, r2t, rows, torch.tensor(seq), torch.tensor(pre),
                                        torch.tensor(ext), D ** -0.5, True, True)
        assert (o.view(-1, Hq, D) - want).abs().max().item() < 1e-5
        dseq = [n + 1 for n in seq]
        dloc = torch.stack([r2t[rows[b], seq[b]] for b in range(3)]).long()
        qd = cases.randn((3, Hq * D), 402, dt)
        fb = _Only(ForwardBatch(forward_mode=mode(ForwardMode.DECODE), batch_size=3, input_ids=torch.zeros(3, dtype=torch.int64),
                                req_pool_indices=rows, seq_lens=torch.tensor(dseq), out_cache_loc=dloc, seq_lens_sum=sum(dseq),
                                seq_lens_cpu=torch.tensor(dseq), req_to_token_pool=sr2t, token_to_kv_pool=spool, attn_backend=be),
                   attrs["forward_batch"])
        be.init_forward_metadata(fb)
        o = be.forward(qd, kc[dloc].clone(), vc[dloc].clone(), layer, fb)
        want = ops_ref.decode_attention(qd.view(3, Hq, D), kc, vc, r2t, rows, torch.tensor(dseq), D ** -0.5, True)
        assert (o.view(3, Hq, D) - want).abs().max().item() < 1e-5
        # the graph hooks as the reference's graph runner calls them (positionally, no extra arguments)
        be.init_cuda_graph_state(4, 4)
        be.init_forward_metadata_capture_cuda_graph(3, 3, rows, torch.tensor(dseq), None, mode(ForwardMode.DECODE), None)
        be.init_forward_metadata_replay_cuda_graph(3, rows, torch.tensor(dseq), sum(dseq), None, mode(ForwardMode.DECODE), None, None)
        assert be.get_cuda_graph_seq_len_fill_value() == 1
    finally:
        ops_backend.set_ops_backend(None)


def test_logits_processor_final_logit_softcapping():
    """LogitsProcessor applies config.final_logit_softcapping to the fp32 logits and treats a negative
    value as off (python/sglang/srt/layers/logits_processor.py:217-224, :492-493); the oracle's
    restatement of the kernel's arithmetic equals cap * tanh(x / cap)."""
    from types import SimpleNamespace

    import cpu_ops
    from oracle import ops_ref
    from sglang_custom_amd import ops_backend
    from sglang_custom_amd.layers.logits_processor import LogitsProcessor
    from sglang_custom_amd.model_executor.forward_batch_info import ForwardMode

    x = torch.linspace(-200, 200, 4001)
    assert (ops_ref.softcap(x, 30.0).double() - 30.0 * torch.tanh(x.double() / 30.0)).abs().max() < 3e-5
    assert torch.isfinite(ops_ref.softcap(torch.tensor([3.0e38, -3.0e38]), 30.0)).all()
    ops_backend.set_ops_backend(cpu_ops)
    try:
        g = torch.Generator().manual_seed(3)
        w, h = torch.randn((50, 16), generator=g), torch.randn((4, 16), generator=g) * 6
        fb = SimpleNamespace(forward_mode=ForwardMode.DECODE)
        out = {c: LogitsProcessor(SimpleNamespace(vocab_size=50, final_logit_softcapping=c))(
            None, h, SimpleNamespace(weight=w), fb).next_token_logits for c in (None, -1.0, 10.0)}
        assert torch.equal(out[None], out[-1.0])
        assert out[None].abs().max() > 10 and out[10.0].abs().max() <= 10
        assert torch.allclose(out[10.0], 10.0 * torch.tanh(out[None] / 10.0), atol=1e-5)
    finally:
        ops_backend.set_ops_backend(None)


@pytest.mark.parametrize("preset", ["tiny-deepseek", "tiny-deepseek-yarn"])
def test_deepseek_dense_stack_host_wiring_matches_oracle(preset):
    """ModelRunner picks DeepseekV2ForCausalLM + MLATokenToKVPool from config.architectures
    (model_config.py:186-204, model_runner.py:950-957, 1192-1206); prefill in the multi-head form, decode
    in the absorbed form; fp32 tokens equal to the oracle stack's and logits within 1e-4.  A
    mixture-of-experts configuration is refused."""
    import cases
    import cpu_ops
    from oracle.deepseek_ref import OracleDeepseek
    from sglang_custom_amd import bench_one_batch as bob
    from sglang_custom_amd import ops_backend
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.managers.schedule_batch import Req
    from sglang_custom_amd.mem_cache.memory_pool import MLATokenToKVPool
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.server_args import ServerArgs

    ops_backend.set_ops_backend(cpu_ops)
    try:
        cfg = hf_config_dict(preset, "float32")
        w = make_synthetic_llama_weights(cfg, seed=1234, dtype=torch.float32)
        sa = ServerArgs(dtype="float32", device="cpu", max_total_tokens=512, max_running_requests=8, disable_cuda_graph=True)
        mr = ModelRunner(ModelConfig(cfg, dtype="float32"), sa, weights=w)
        assert isinstance(mr.token_to_kv_pool, MLATokenToKVPool)
        prompts = cases.e2e_prompts(cfg["vocab_size"], (5, 33))
        ref_toks, ref_logits = OracleDeepseek(cfg, w, torch.float32, max_total_tokens=512, max_reqs=8,
                                              context_len=256).generate_greedy(prompts, 4)
        nxt, lg, batch = bob.extend([Req(i, p) for i, p in enumerate(prompts)], mr)
        toks, err = [nxt.clone()], (lg - ref_logits[0]).abs().max().item()
        for s in range(3):
            nxt, lg = bob.decode(nxt, batch, mr)
            toks.append(nxt.clone())
            err = max(err, (lg - ref_logits[s + 1]).abs().max().item())
        assert torch.equal(torch.stack(toks, 1), ref_toks) and err < 1e-4, err
        moe = dict(cfg, n_routed_experts=8, first_k_dense_replace=1)
        with pytest.raises(NotImplementedError):
            ModelRunner(ModelConfig(moe, dtype="float32"), sa, weights=w)
    finally:
        ops_backend.set_ops_backend(None)


def test_input_token_logprobs_match_reference_vectors():
    """Prefill with return_logprob: ScheduleBatch.prepare_for_extend's bookkeeping (extend_logprob_start_lens,
    the next-token ids of every kept row, zero padding) and LogitsProcessor's extend_return_logprob branch
    (row selection, log-softmax, per-request top-k) against vectors recorded from the reference's own
    classes on CPU (tests/golden/logprob.json): plain prefill and an extend over a cached prefix, with
    logprob_start_len before / inside / past the new tokens."""
    import json

    import cases

    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "logprob.json")))
    cfg, w, dt, mr = _build(golden["preset"], "float32")
    cases.check_logprob_records(cases.run_logprob_scenarios(mr, golden), golden, 1e-4)
