"""Generate tests/golden/cache.json from the REFERENCE's own code (build container only):
  * paged slot allocation: the torch restatements the reference keeps beside its Triton kernels
    (python/sglang/srt/mem_cache/allocator.py alloc_extend_kernel_ascend / alloc_decode_kernel_ascend)
    on seeded batches;
  * radix prefix tree: a seeded trace of match_prefix / insert / lock / evict operations run through
    the reference's RadixCache (mem_cache/radix_cache.py, page_size 1) with the observable results
    after each operation.
Inputs and outputs only; no reference source is written anywhere.
    python tests/golden/gen_cache_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import _ref_harness  # noqa: E402

_ref_harness.install()

import torch  # noqa: E402

import sglang.srt.managers.schedule_batch  # noqa: E402,F401  (import order: avoids a circular import)
from sglang.srt.mem_cache.allocator import alloc_decode_kernel_ascend, alloc_extend_kernel_ascend  # noqa: E402
from sglang.srt.mem_cache.radix_cache import RadixCache  # noqa: E402


def paged_cases():
    out = []
    rng = random.Random(7)
    for page in (2, 4, 16, 64):
        for _ in range(6):
            bs = rng.randint(1, 9)
            pre, seq, last = [], [], []
            next_page = 1000
            for _b in range(bs):
                p = rng.choice([0, 0, rng.randint(1, 3 * page), page, 2 * page])
                s = p + rng.randint(1, 3 * page + 3)
                # the torch restatement is only defined when the extension reaches the end of the
                # request's last page (its Triton twin returns early otherwise): keep to that domain
                s = max(s, (p + page - 1) // page * page)
                if s == p:
                    s += page
                pre.append(p)
                seq.append(s)
                # the slot of the last prefix token sits at offset (p - 1) % page of some page
                last.append(-1 if p == 0 else next_page * page + (p - 1) % page)
                next_page += 1
            need = sum((s + page - 1) // page - (p + page - 1) // page for p, s in zip(pre, seq))
            free = rng.sample(range(1, 500), need + 3)
            ext = sum(s - p for p, s in zip(pre, seq))
            o = torch.zeros(ext, dtype=torch.int64)
            alloc_extend_kernel_ascend(torch.tensor(pre), torch.tensor(seq), torch.tensor(last), torch.tensor(free), o, page, "cpu")
            case = dict(page=page, prefix_lens=pre, seq_lens=seq, last_loc=last, free_pages=free, extend_out=o.tolist())
            # decode right after: one more token per request; last_loc = its last slot
            last2, off = [], 0
            for p, s in zip(pre, seq):
                off += s - p
                last2.append(int(o[off - 1]))
            seq2 = [s + 1 for s in seq]
            need2 = sum(1 for s in seq2 if (s - 1) % page == 0)
            free2 = rng.sample(range(500, 900), need2 + 2)
            o2 = torch.zeros(bs, dtype=torch.int64)
            alloc_decode_kernel_ascend(torch.tensor(seq2), torch.tensor(last2), torch.tensor(free2), o2, page)
            case.update(decode_seq_lens=seq2, decode_last_loc=last2, decode_free_pages=free2, decode_out=o2.tolist())
            out.append(case)
    return out


class _Alloc:
    def __init__(self):
        self.freed = []
        self.device = "cpu"

    def free(self, idx):
        self.freed.append(sorted(int(v) for v in idx.tolist()))


def radix_trace(page_size=1):
    """page_size > 1: stems and tails are longer (whole pages must repeat for anything to match), the
    alphabet is smaller, and inserted keys are cut to whole pages as cache_finished_req does."""
    rng = random.Random(11 if page_size == 1 else 29)
    alloc = _Alloc()
    tree = RadixCache(req_to_token_pool=None, token_to_kv_pool_allocator=alloc, page_size=page_size)
    ops, next_slot, locked = [], 1 if page_size == 1 else page_size, []
    hi, stem_len, tail = (5, 12, 6) if page_size == 1 else (2, 12 * page_size, 3 * page_size)
    stems = [[rng.randint(0, hi) for _ in range(rng.randint(3, stem_len))] for _ in range(5)]
    for step in range(160):
        kind = rng.choice(["insert", "insert", "match", "match", "lock", "unlock", "evict"])
        key = list(rng.choice(stems))[: rng.randint(1, stem_len)] + [rng.randint(0, hi) for _ in range(rng.randint(0, tail))]
        rec = dict(op=kind)
        if kind == "insert" and page_size > 1:
            key = key[: len(key) // page_size * page_size]
            if not key:
                continue
        if kind == "insert":
            val = list(range(next_slot, next_slot + len(key)))
            next_slot += len(key)
            rec.update(key=key, value=val, ret=int(tree.insert(key, torch.tensor(val, dtype=torch.int64))))
        elif kind == "match":
            res = tree.match_prefix(key)
            rec.update(key=key, matched=[int(v) for v in res.device_indices.tolist()])
        elif kind == "lock":
            res = tree.match_prefix(key)
            tree.inc_lock_ref(res.last_device_node)
            locked.append(key)
            rec.update(key=key)
        elif kind == "unlock":
            if not locked:
                continue
            k = locked.pop(rng.randrange(len(locked)))
            res = tree.match_prefix(k)
            tree.dec_lock_ref(res.last_device_node)
            rec.update(key=k)
        else:
            n = rng.randint(1, 10)
            before = len(alloc.freed)
            tree.evict(n)
            rec.update(num_tokens=n, freed=sorted(v for grp in alloc.freed[before:] for v in grp))
        rec.update(total=int(tree.total_size()), evictable=int(tree.evictable_size()), protected=int(tree.protected_size()))
        ops.append(rec)
    return ops


def main():
    out = dict(paged=paged_cases(), radix=radix_trace(), radix_page4=radix_trace(4))
    path = os.path.join(HERE, "cache.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["paged"]), "paged cases,", len(out["radix"]), "radix ops")


if __name__ == "__main__":
    main()
