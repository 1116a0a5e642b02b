"""Packed decode GEMM in deferred-reduction mode (fp32 partials out): sweep tiles per workgroup x splits."""
import sys, torch
sys.path.insert(0, '.')
from sglang_custom_amd import _lib, ops
_lib.use_dev_library()   # ablation switches and probes live in the development build only
lib = _lib.load()
def graph_time(fn, n=40, reps=3):
    fn(0); fn(1); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, N, K in (("qkv", 6144, 4096), ("o", 4096, 4096), ("down", 4096, 14336)):
    nw = 8
    ws = [ops.pack_weight(torch.randn(N, K, device='cuda', dtype=torch.bfloat16) * 0.02) for _ in range(nw)]
    x = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    lib.mi355_debug_set(2, 0); lib.mi355_debug_set(4, 0)
    print(f"{name} plan: {graph_time(lambda i: ops.linear_packed_partials(x, ws[i % nw], N)):6.2f} us (splits {lib.mi355_linear_packed_splits(M, N, K)})", flush=True)
    for waves in (4, 6, 7, 8):
        row = []
        for fs in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16):
            if fs > K // 512: continue
            lib.mi355_debug_set(4, waves); lib.mi355_debug_set(2, fs)
            if lib.mi355_linear_packed_splits(M, N, K) != fs: continue
            row.append(f"s{fs}:{graph_time(lambda i: ops.linear_packed_partials(x, ws[i % nw], N)):5.1f}")
        print(f"{name} waves={waves}  " + " ".join(row), flush=True)
    lib.mi355_debug_set(2, 0); lib.mi355_debug_set(4, 0)
