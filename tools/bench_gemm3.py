"""Dev tool: packed decode GEMM, sweep tiles-per-workgroup (waves) x split-K per shape."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from sglang_custom_amd import ops, _lib
_lib.use_dev_library()   # ablation switches and probes live in the development build only
lib = _lib.load()
lib.mi355_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]
def timeit(fn, reps=20):
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
for name, N, K in shapes:
    w = [torch.randn(N, K, device='cuda').bfloat16() for _ in range(6)]
    wp = [ops.pack_weight(t) for t in w]
    x = torch.randn(M, K, device='cuda').bfloat16()
    lib.mi355_debug_set(0, 1); lib.mi355_debug_set(2, 0); lib.mi355_debug_set(4, 0)
    ref = ops.linear(x, w[0])
    lib.mi355_debug_set(0, 0)
    t = timeit(lambda i: ops.linear_packed(x, wp[i % 6], N))
    ok = torch.equal(ops.linear_packed(x, wp[0], N), ref)
    print(f"{name:8s} PLAN: {t:6.1f} us {N * K * 2 / t / 1e6:5.2f} TB/s bit-equal={ok}", flush=True)
    for waves in (4, 6, 7, 8):
        row = [f"{name:8s} waves={waves}:"]
        for fs in (1, 2, 3, 4, 6, 8):
            if fs > K // 512: continue
            lib.mi355_debug_set(4, waves); lib.mi355_debug_set(2, fs)
            got = ops.linear_packed(x, wp[0], N)
            d = (got.float() - ref.float()).abs().max().item()
            t = timeit(lambda i: ops.linear_packed(x, wp[i % 6], N))
            row.append(f"s{fs} {t:5.1f}us{'' if d < 1.01 else ' BAD d=%.3g' % d} |")
        print(" ".join(row), flush=True)
    lib.mi355_debug_set(2, 0); lib.mi355_debug_set(4, 0)
    del w, wp
