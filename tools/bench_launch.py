"""Launch-floor and cache-retention probes (development aid).
1. Per-kernel cost of a serial chain of empty kernels inside one HIP graph.
2. Re-reading the same buffer in back-to-back kernels: does L2 / MALL keep it across a kernel boundary?"""
import sys, torch
sys.path.insert(0, '.')
from sglang_custom_amd import _lib
_lib.use_dev_library()   # ablation switches and probes live in the development build only
lib = _lib.load()
sink = torch.zeros(4, dtype=torch.int32, device='cuda')
def graph_time(fns, reps=3):
    for f in fns[:2]: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns: f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3
def rd(t, grid, nbytes=None):
    nb = t.numel() * 4 if nbytes is None else nbytes
    assert nb <= t.numel() * 4
    return lambda: lib.mi355_debug_read_bw(t.data_ptr(), nb, grid, 8, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
small = torch.zeros(1 << 20, dtype=torch.int32, device='cuda')
for grid in (1, 256, 1024):
    print(f"empty chain grid {grid:5d}: {graph_time([rd(small, grid, 0)] * 200):6.2f} us/kernel", flush=True)
for mb in (4, 8, 16, 24, 32, 64, 128, 192, 256, 512):
    n = mb * 262144
    same = torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, device='cuda')
    many = [torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, device='cuda') for _ in range(max(2, 2048 // mb))]
    for grid in (256, 1024):
        t_same = graph_time([rd(same, grid)] * 40)
        t_diff = graph_time([rd(many[i % len(many)], grid) for i in range(40)])
        print(f"{mb:4d} MB grid {grid:5d}: same buffer {t_same:7.2f} us ({mb * 1.048576 / t_same:5.2f} TB/s)  "
              f"rotating buffers {t_diff:7.2f} us ({mb * 1.048576 / t_diff:5.2f} TB/s)", flush=True)
    del same, many
