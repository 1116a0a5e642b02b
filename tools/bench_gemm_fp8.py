"""Dev tool: fp8 W8A8 packed decode GEMM (deferred split-K partials, as the decode step calls it), sweep
tiles-per-workgroup (waves) x split-K per Llama-3-8B shape; the PLAN row is what plan_fp8 picks."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from sglang_custom_amd import ops, _lib
_lib.use_dev_library()   # ablation switches live in the development build only
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
    w = [(torch.randn(N, K, device='cuda') * 30).to(torch.float8_e4m3fn) for _ in range(6)]
    wp = [ops.pack_weight_fp8(t) for t in w]
    xq = (torch.randn(M, K, device='cuda') * 30).to(torch.float8_e4m3fn)
    xs = torch.rand(M, 1, device='cuda') + 0.5
    wsc = torch.rand(N, 1, device='cuda') + 0.5
    ref = (xq.float() @ w[0].float().t())
    def run(i):
        if name == "gate_up":   # the step calls the in-kernel epilogue here (wide output, never split)
            return ops.fp8_scaled_mm_packed(xq, wp[i % 6], N, xs, wsc, torch.bfloat16)
        return ops.fp8_packed_partials(xq, wp[i % 6], N)
    def check():
        if name == "gate_up":
            return (run(0).float() - (ref * xs) * wsc.view(1, -1)).abs().max().item() / ref.abs().max().item()
        return (run(0).sum(0) - ref).abs().max().item() / ref.abs().max().item()
    lib.mi355_debug_set(2, 0); lib.mi355_debug_set(4, 0)
    t = timeit(run)
    print(f"{name:8s} PLAN: {t:6.1f} us {N * K / t / 1e6:5.2f} TB/s relerr={check():.2e} splits={lib.mi355_fp8_packed_splits(M, N, K)}", flush=True)
    for waves in (4, 6, 7, 8):
        row = [f"{name:8s} waves={waves}:"]
        for fs in (1, 2, 4, 7, 8, 14):
            if fs > K // 1024 or (K // 512) % fs: continue
            lib.mi355_debug_set(4, waves); lib.mi355_debug_set(2, fs)
            d = check()
            t = timeit(run)
            row.append(f"s{fs} {t:5.1f}us{'' if d < 2e-2 else ' BAD d=%.3g' % d} |")
        print(" ".join(row), flush=True)
    lib.mi355_debug_set(2, 0); lib.mi355_debug_set(4, 0)
    del w, wp
