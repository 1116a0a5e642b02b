"""Dev tool: greedy argmax over fp32 logits [rows, 128256]: one workgroup per row (mi355_argmax) against
several workgroups per row meeting in an atomic-max cell (mi355_argmax_chunked, what ops.argmax runs for
fewer than 256 rows).  Rows on the command line."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops, _lib
lib = _lib.load()
V = 128256
def timeit(fn):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(16): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 48 * 1e3
for rows in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64]:
    lg = [torch.randn(rows, V, device='cuda') for _ in range(4)]
    got = ops.argmax(lg[0])
    ok = torch.equal(got.cpu(), lg[0].cpu().argmax(-1))
    ids = torch.empty(rows, dtype=torch.int64, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    t_one = timeit(lambda i: lib.mi355_argmax(ids.data_ptr(), lg[i % 4].data_ptr(), rows, V, V, torch.cuda.current_stream().cuda_stream))
    t_chunk = timeit(lambda i: ops.argmax(lg[i % 4]))
    print(f"rows={rows}: one workgroup per row {t_one:6.2f} us   chunked {t_chunk:6.2f} us  equal={ok}", flush=True)
