"""Micro-benchmark of the decode projection kernel per shape and variant (dev tool)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from sglang_custom_amd import ops, _lib
lib = _lib.load()
lib.mi355_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
dev = 'cuda'
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336), ("lm_head", 128256, 4096)]
NB = 6  # distinct weight buffers so that nothing is served from L2 / Infinity Cache
def timeit(fn, reps=20):
    """GPU time per call with the calls replayed from a HIP graph (no host launch gaps)."""
    fn(0); fn(1); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3
# HBM copy reference
a = torch.empty(1 << 30, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
t = timeit(lambda i: b.copy_(a), 10)
print(f"copy 1GiB: {t:.1f} us -> {2 * (1 << 30) / t / 1e6:.2f} TB/s (read+write)")
t = timeit(lambda i: a.sum(dtype=torch.int64) if False else torch.sum(a.view(torch.int32)), 10)
print(f"sum-read 1GiB: {t:.1f} us -> {(1 << 30) / t / 1e6:.2f} TB/s")
del a, b
for name, N, K in shapes:
    nb = NB if N * K < 2e8 else 2
    ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(nb)]
    x = torch.randn(M, K, device=dev).bfloat16()
    row = [f"{name:8s} {N:6d}x{K:5d}  ideal@6.3TB/s {N * K * 2 / 6.3e6:6.1f} us |"]
    for variant, label in ((1, "v1"), (0, "stream"), (2, "stream-noX")):
        lib.mi355_debug_set(0, variant)
        t = timeit(lambda i: ops.linear(x, ws[i % nb]))
        row.append(f"{label} {t:6.1f} us {N * K * 2 / t / 1e6:5.2f} TB/s |")
    wps = [ops.pack_weight(w) for w in ws]
    lib.mi355_debug_set(0, 0)
    ref = ops.linear(x, ws[0]); got = ops.linear_packed(x, wps[0], N)
    assert torch.equal(ref, got), (ref - got).abs().max()
    for variant, label in ((1, "v1-packed"), (0, "stream-packed"), (2, "stream-packed-noX")):
        lib.mi355_debug_set(0, variant)
        t = timeit(lambda i: ops.linear_packed(x, wps[i % nb], N))
        row.append(f"{label} {t:6.1f} us {N * K * 2 / t / 1e6:5.2f} TB/s |")
    del wps
    lib.mi355_debug_set(0, 0)
    t = timeit(lambda i: torch.nn.functional.linear(x, ws[i % nb]))
    row.append(f"hipblaslt {t:6.1f} us {N * K * 2 / t / 1e6:5.2f} TB/s")
    print(" ".join(row), flush=True)
    del ws
