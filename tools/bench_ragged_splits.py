"""Decode attention over a ragged batch: every sequence cut into num_kv_splits (uniform plan) against the
per-sequence plan (kv_chunk_keys: ceil(len / chunk) splits per sequence, chunk = ceil(longest / num_kv_splits))."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops
D, Hq, Hkv = 128, 32, 8
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3
gen = torch.Generator().manual_seed(0)
for name, lens in (("8 x 16..4096 ragged", [16, 4096, 64, 300, 1000, 33, 2048, 128]),
                   ("8 x 4096 uniform", [4096] * 8),
                   ("16 x ragged", torch.randint(16, 4097, (16,), generator=gen).tolist()),
                   ("64 x ragged (one 4096, rest <= 256)", [4096] + torch.randint(16, 257, (63,), generator=gen).tolist())):
    B = len(lens)
    total = sum(lens)
    kc = torch.randn(total + 1, Hkv, D, device='cuda').bfloat16()
    vc = torch.randn(total + 1, Hkv, D, device='cuda').bfloat16()
    r2t = torch.zeros(B, max(lens), dtype=torch.int32, device='cuda')
    p = 1
    for b, n in enumerate(lens):
        r2t[b, :n] = torch.arange(p, p + n, dtype=torch.int32, device='cuda'); p += n
    rpi = torch.arange(B, dtype=torch.int64, device='cuda')
    lt = torch.tensor(lens, dtype=torch.int64, device='cuda')
    q = torch.randn(B, Hq, D, device='cuda').bfloat16()
    o = torch.empty(B, Hq, D, device='cuda', dtype=torch.bfloat16)
    for S in sorted({ops.decode_num_splits(B, Hkv, max(lens), 64), ops.decode_num_splits(B, Hkv, 8192, 64)}):
        if S == 1: continue
        al = torch.empty(B * Hq * S * D, dtype=torch.float32, device='cuda'); ll = torch.empty(B * Hq * S, dtype=torch.float32, device='cuda')
        tk = torch.zeros(B * Hq, dtype=torch.int32, device='cuda')
        ck = torch.tensor([(max(lens) + S - 1) // S], dtype=torch.int32, device='cuda')
        t0 = timeit(lambda: ops.decode_attention_paged(q, kc, vc, o, r2t, rpi, lt, al, ll, S, D ** -0.5, tickets=tk))
        t1 = timeit(lambda: ops.decode_attention_paged(q, kc, vc, o, r2t, rpi, lt, al, ll, S, D ** -0.5, tickets=tk, kv_chunk_keys=ck))
        print(f"{name:40s} splits={S:2d}: uniform {t0:6.1f} us   per-sequence {t1:6.1f} us", flush=True)
