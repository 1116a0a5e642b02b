"""Dev tool: greedy argmax over a [64, 128256] fp32 logits matrix, replayed from a graph."""
import sys, torch
sys.path.insert(0, '.')
from sglang_custom_amd import ops
x = [torch.randn(64, 128256, device='cuda') for _ in range(6)]
ops.argmax(x[0]); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(30): ops.argmax(x[i % 6])
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
print("argmax 64x128256: %.2f us" % (e0.elapsed_time(e1) / 60 * 1e3))
assert torch.equal(ops.argmax(x[1]).cpu(), x[1].argmax(-1).cpu())
