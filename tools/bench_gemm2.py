"""Dev tool: packed / unpacked decode GEMM, ring vs two-launch kernel, split sweep."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from sglang_custom_amd import ops, _lib
lib = _lib.load()
lib.mi355_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336), ("lm_head", 128256, 4096)]
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
    nb = 6 if N * K < 2e8 else 2
    w = [torch.randn(N, K, device='cuda').bfloat16() for _ in range(nb)]
    wp = [ops.pack_weight(t) for t in w]
    x = torch.randn(M, K, device='cuda').bfloat16()
    lib.mi355_debug_set(0, 1); lib.mi355_debug_set(2, 0)
    ref = ops.linear(x, w[0])
    for fs in (0, 1, 2, 4, 8):
        if fs and N > 60000: continue
        row = [f"{name:8s} splits={fs or 'plan'}:"]
        lib.mi355_debug_set(2, fs)
        for variant, label in ((1, "two-launch"), (0, "ring")):
            lib.mi355_debug_set(0, variant)
            got = ops.linear_packed(x, wp[0], N); got2 = ops.linear(x, w[0])
            d1 = (got.float() - ref.float()).abs().max().item(); d2 = (got2.float() - ref.float()).abs().max().item()
            t = timeit(lambda i: ops.linear_packed(x, wp[i % nb], N))
            t2 = timeit(lambda i: ops.linear(x, w[i % nb]))
            row.append(f"{label}: packed {t:6.1f} us {N * K * 2 / t / 1e6:5.2f} TB/s (d={d1:.3g}) unpacked {t2:6.1f} us (d={d2:.3g}) |")
        print(" ".join(row), flush=True)
    lib.mi355_debug_set(0, 0); lib.mi355_debug_set(2, 0)
    got = ops.linear_packed(x, wp[0], N)
    print(f"   plan/ring bit-equal to two-launch plan: {torch.equal(got, ref)}")
    del w, wp
