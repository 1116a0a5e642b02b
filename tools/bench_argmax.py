"""Dev tool: greedy argmax over fp32 logits [rows, 128256], rows given on the command line."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops
V = 128256
for rows in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64]:
    lg = [torch.randn(rows, V, device='cuda') for _ in range(4)]
    got = ops.argmax(lg[0])
    ok = torch.equal(got.cpu(), lg[0].cpu().argmax(-1))
    g = torch.cuda.CUDAGraph()
    ops.argmax(lg[1]); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for i in range(16): ops.argmax(lg[i % 4])
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"rows={rows}: {e0.elapsed_time(e1) / 48 * 1e3:6.2f} us  equal={ok}", flush=True)
