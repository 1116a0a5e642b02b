"""Dev tool: the lm_head GEMM (fp32 logits) as planned, per batch size given on the command line."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops, _lib
_lib.use_dev_library()
lib = _lib.load()
lib.mi355_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
N, K = 128256, 4096
def timeit(fn, reps=6):
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
w = [torch.randn(N, K, device='cuda').bfloat16() for _ in range(3)]
wp = [ops.pack_weight(t) for t in w]
for M in [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32, 64]:
    x = torch.randn(M, K, device='cuda').bfloat16()
    got = ops.linear_packed(x, wp[0], N, out_dtype=torch.float32)
    ref = ops.linear(x, w[0], out_dtype=torch.float32)
    t = timeit(lambda i: ops.linear_packed(x, wp[i % 3], N, out_dtype=torch.float32))
    print(f"M={M}: {t:6.1f} us {N * K * 2 / t / 1e6:5.2f} TB/s bit-equal={torch.equal(got, ref)}", flush=True)
