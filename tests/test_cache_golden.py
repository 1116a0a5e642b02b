"""Paged slot allocation and the radix prefix tree against fixtures produced by the reference's own
code (tests/golden/cache.json, made by tests/golden/gen_cache_golden.py):
  * oracle.alloc_extend_pages / alloc_decode_pages == the reference's torch restatements of its
    allocation kernels, case by case; on the GPU the HIP kernels equal the oracle bit for bit, also
    on the cases outside the fixture's domain (extensions that stay inside the old page);
  * sglang_custom_amd.mem_cache.RadixCache replays the reference RadixCache's trace of match /
    insert / lock / unlock / evict with identical observable results after every operation;
  * the paged allocator keeps requests page-aligned through a scheduler run whose outputs equal the
    oracle's for each prompt alone."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import ops_ref

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "cache.json")) as f:
    GOLD = json.load(f)


def test_paged_allocation_oracle_matches_reference():
    for c in GOLD["paged"]:
        out, used = ops_ref.alloc_extend_pages(c["prefix_lens"], c["seq_lens"], c["last_loc"], c["free_pages"], c["page"])
        assert out.tolist() == c["extend_out"], c
        page = c["page"]
        assert used == sum((s + page - 1) // page - (p + page - 1) // page for p, s in zip(c["prefix_lens"], c["seq_lens"]))
        out2, used2 = ops_ref.alloc_decode_pages(c["decode_seq_lens"], c["decode_last_loc"], c["decode_free_pages"], page)
        assert out2.tolist() == c["decode_out"], c
        assert used2 == sum(1 for s in c["decode_seq_lens"] if (s - 1) % page == 0)


def _random_paged_case(rng, page, bs):
    pre, seq, last = [], [], []
    for b in range(bs):
        p = rng.choice([0, rng.randint(1, 3 * page), page])
        s = p + rng.randint(1, 3 * page + 3)     # includes extensions that stay inside the old page
        pre.append(p)
        seq.append(s)
        last.append(-1 if p == 0 else (2000 + b) * page + (p - 1) % page)
    need = sum((s + page - 1) // page - (p + page - 1) // page for p, s in zip(pre, seq))
    free = rng.sample(range(1, 1500), need + 2)
    return pre, seq, last, free


def test_paged_allocation_invariants():
    """Every request's slots are consecutive within pages, continue its last page, never collide."""
    rng = random.Random(3)
    for page in (2, 16):
        for _ in range(20):
            pre, seq, last, free = _random_paged_case(rng, page, rng.randint(1, 12))
            out, _ = ops_ref.alloc_extend_pages(pre, seq, last, free, page)
            out = out.tolist()
            assert len(set(out)) == len(out)
            off = 0
            for p, s, ll in zip(pre, seq, last):
                mine = out[off: off + s - p]
                off += s - p
                for i, slot in enumerate(mine):
                    assert slot % page == (p + i) % page            # position within the page = position in the sequence
                    if i > 0 and (p + i) % page != 0:
                        assert slot == mine[i - 1] + 1
                if p % page != 0:
                    assert mine[0] == ll + 1


@pytest.mark.gpu
def test_paged_allocation_kernels_match_oracle():
    from sglang_custom_amd import ops

    rng = random.Random(5)
    dev = "cuda"
    for page in (2, 4, 16, 64):
        for bs in (1, 7, 64, 300):
            pre, seq, last, free = _random_paged_case(rng, page, bs)
            ref, used = ops_ref.alloc_extend_pages(pre, seq, last, free, page)
            tt = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
            out, used_gpu = ops.alloc_extend_pages(tt(pre), tt(seq), tt(last), tt(free), page, ref.numel())
            assert used_gpu == used and torch.equal(out.cpu(), ref)
            seq2 = [s + 1 for s in seq]
            last2, off = [], 0
            for p, s in zip(pre, seq):
                off += s - p
                last2.append(int(ref[off - 1]))
            free2 = rng.sample(range(3000, 5000), bs + 1)
            ref2, used2 = ops_ref.alloc_decode_pages(seq2, last2, free2, page)
            out2, used2_gpu = ops.alloc_decode_pages(tt(seq2), tt(last2), tt(free2), page)
            assert used2_gpu == used2 and torch.equal(out2.cpu(), ref2)


@pytest.mark.parametrize("name,page_size", [("radix", 1), ("radix_page4", 4)])
def test_radix_tree_replays_reference_trace(name, page_size):
    from sglang_custom_amd.mem_cache.radix_cache import RadixCache

    class Alloc:
        device = "cpu"

        def __init__(self):
            self.freed = []

        def free(self, idx):
            self.freed.extend(int(v) for v in idx.tolist())

    alloc = Alloc()
    tree = RadixCache(req_to_token_pool=None, token_to_kv_pool_allocator=alloc, page_size=page_size)
    for i, rec in enumerate(GOLD[name]):
        op = rec["op"]
        if op == "insert":
            assert tree.insert(rec["key"], torch.tensor(rec["value"], dtype=torch.int64)) == rec["ret"], (i, rec)
        elif op == "match":
            assert tree.match_prefix(rec["key"])[0].tolist() == rec["matched"], (i, rec)
        elif op == "lock":
            tree.inc_lock_ref(tree.match_prefix(rec["key"])[1])
        elif op == "unlock":
            tree.dec_lock_ref(tree.match_prefix(rec["key"])[1])
        else:
            before = len(alloc.freed)
            tree.evict(rec["num_tokens"])
            assert sorted(alloc.freed[before:]) == rec["freed"], (i, rec)
        assert (tree.total_size(), tree.evictable_size(), tree.protected_size()) == \
            (rec["total"], rec["evictable"], rec["protected"]), (i, rec)


def _scheduler_run(ops_backend_module, device, page, dtype="float32", weights_to=None, graphs=False, radix=False):
    from oracle.llama_ref import OracleLlama
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.managers.schedule_batch import Req
    from sglang_custom_amd.managers.scheduler import Scheduler
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.sampling.sampling_params import SamplingParams
    from sglang_custom_amd.server_args import ServerArgs

    cfg = hf_config_dict("tiny-llama-d64", dtype)
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=torch.float32)
    sa = ServerArgs(dtype=dtype, device=device, max_total_tokens=320, max_running_requests=4,
                    disable_cuda_graph=not graphs, cuda_graph_bs=[1, 2, 4], chunked_prefill_size=24,
                    max_prefill_tokens=64, page_size=page)
    mr = ModelRunner(ModelConfig(cfg, dtype=dtype), sa,
                     weights={k: (v.to(weights_to) if weights_to else v) for k, v in w.items()})
    rng = np.random.RandomState(9)
    spec = [(5, 6), (41, 3), (17, 9), (33, 2), (70, 4), (9, 12), (24, 5)]
    reqs = [Req(i, rng.randint(0, cfg["vocab_size"], n).tolist(), SamplingParams(max_new_tokens=m, temperature=0))
            for i, (n, m) in enumerate(spec)]
    if radix:   # prompts that share whole pages with earlier ones (and a few tokens past a page boundary)
        stem = rng.randint(0, cfg["vocab_size"], 2 * page + 3).tolist()
        for r in reqs[1:]:
            if len(r.origin_input_ids) > len(stem):
                r.origin_input_ids[: len(stem)] = stem
                r.fill_ids = r.origin_input_ids
    sch = Scheduler(mr, enable_radix_cache=radix)
    for r in reqs[:3]:
        sch.add_request(r)
    arrivals = {2: reqs[3], 4: reqs[4], 5: reqs[5], 9: reqs[6]}
    it = 0
    table = mr.req_to_token_pool.req_to_token
    while sch.waiting_queue or sch.chunked_req is not None or not sch.running_batch.is_empty() or arrivals:
        if it in arrivals:
            sch.add_request(arrivals.pop(it))
        assert sch.step() or arrivals
        # every running request: slot of token t sits at offset t % page of its page, pages are private
        seen = set()
        for r in sch.running_batch.reqs:
            n = len(r.origin_input_ids) + len(r.output_ids) - 1
            row = table[r.req_pool_idx, :n].tolist()
            assert all(s % page == t % page for t, s in enumerate(row)), (r.rid, row)
            pages = {s // page for s in row}
            if radix:   # whole pages of a cached prefix are shared; the page being filled never is
                pages = {row[-1] // page} if n % page else set()
            assert not (pages & seen)
            seen |= pages
        it += 1
        assert it < 500
    assert len(sch.finished) == len(reqs)
    alloc = mr.token_to_kv_pool_allocator
    assert alloc.available_size() + sch.tree_cache.evictable_size() == mr.max_total_num_tokens
    if radix:
        assert sch.cached_prefix_tokens >= 2 * page and sch.cached_prefix_tokens % page == 0   # whole pages were reused
        assert sch.tree_cache.protected_size() == 0 and sch.tree_cache.total_size() % page == 0
        sch.tree_cache.evict(1 << 30)
        assert alloc.available_size() == mr.max_total_num_tokens
    oracle = OracleLlama(cfg, w, torch.float32, max_total_tokens=512)
    for r, (n, m) in zip(reqs, spec):
        ref_toks, _ = oracle.generate_greedy([r.origin_input_ids], m)
        assert r.output_ids == ref_toks[0].tolist(), (r.rid, page)


@pytest.mark.parametrize("page", [4, 16])
def test_scheduler_with_paged_allocator_on_cpu(page):
    import cpu_ops
    from sglang_custom_amd import ops_backend

    ops_backend.set_ops_backend(cpu_ops)
    _scheduler_run(cpu_ops, "cpu", page)


@pytest.mark.parametrize("page", [4, 16])
def test_scheduler_with_prefix_tree_over_pages_on_cpu(page):
    import cpu_ops
    from sglang_custom_amd import ops_backend

    ops_backend.set_ops_backend(cpu_ops)
    _scheduler_run(cpu_ops, "cpu", page, radix=True)


@pytest.mark.gpu
@pytest.mark.parametrize("page,graphs,radix", [(16, False, False), (4, True, False), (16, True, True), (4, False, True)])
def test_scheduler_with_paged_allocator_on_gpu(page, graphs, radix):
    from sglang_custom_amd import ops, ops_backend

    ops_backend.set_ops_backend(ops)
    _scheduler_run(ops, "cuda", page, weights_to="cuda", graphs=graphs, radix=radix)
