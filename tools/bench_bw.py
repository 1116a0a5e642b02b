import sys, torch
sys.path.insert(0, '.')
from sglang_custom_amd import _lib
_lib.use_dev_library()   # ablation switches and probes live in the development build only
lib = _lib.load()
dev = 'cuda'
sink = torch.zeros(4, dtype=torch.int32, device=dev)
bufs = {mb: [torch.randint(0, 2**31 - 1, (mb * 262144,), dtype=torch.int32, device=dev) for _ in range(max(2, 1024 // mb))] for mb in (32, 117, 235, 1024)}
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=12):
    fn(0); fn(1); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * reps) * 1e3
for mb, bl in bufs.items():
    for grid in (256, 512, 1024, 2048, 4096):
        for unroll in (4, 8, 16):
            nbytes = bl[0].numel() * 4
            t = timeit(lambda i: lib.mi355_debug_read_bw(bl[i % len(bl)].data_ptr(), nbytes, grid, unroll, sink.data_ptr(), torch.cuda.current_stream().cuda_stream))
            print(f"{mb:5d} MB grid {grid:5d} unroll {unroll:2d}: {t:7.1f} us {nbytes / t / 1e6:5.2f} TB/s", flush=True)
