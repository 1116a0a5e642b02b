"""Dev tool: the 16-bit packed decode GEMM at given (name, N, K) shapes, sweeping tiles-per-workgroup (waves) x
split-K against what plan_skinny picks.  mode: "merge" = split-K summed inside the launch (linear_packed),
"partials" = left to the consumer (linear_packed_partials), "silu" = gate_up with SiLU-and-mul in the epilogue.
usage: python tools/bench_gemm_plan.py M name:N:K:mode [...]   (default: DeepSeek-V3 TP=8 shard shapes)"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from sglang_custom_amd import ops, _lib
_lib.use_dev_library()   # ablation switches live in the development build only
lib = _lib.load()
lib.mi355_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [a.split(":") for a in sys.argv[2:]] or [
    ("qkv_a", 2112, 7168, "partials"), ("q_b", 3072, 1536, "merge"), ("o", 7168, 2048, "merge"),
    ("gate_up", 4608, 7168, "silu"), ("down", 7168, 2304, "merge")]
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
for name, N, K, mode in shapes:
    N, K = int(N), int(K)
    nw = max(2, min(8, (600 << 20) // (N * K * 2)))   # rotate over > 256 MB of weights: no L2 / MALL reuse
    w = [torch.randn(N, K, device='cuda', dtype=torch.bfloat16) * K ** -0.5 for _ in range(nw)]
    wp = [ops.pack_weight_gate_up(t) if mode == "silu" else ops.pack_weight(t) for t in w]
    x = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    ref = x.float() @ w[0].float().t()
    if mode == "silu":
        ref = torch.nn.functional.silu(ref[:, : N // 2]) * ref[:, N // 2:]
    def run(i):
        if mode == "silu": return ops.linear_packed_silu_mul(x, wp[i % nw], N)
        if mode == "partials": return ops.linear_packed_partials(x, wp[i % nw], N)
        return ops.linear_packed(x, wp[i % nw], N)
    def check():
        y = run(0)
        y = y.sum(0) if mode == "partials" else y.float()
        return (y - ref).abs().max().item() / ref.abs().max().item()
    lib.mi355_debug_set(2, 0); lib.mi355_debug_set(4, 0)
    t = timeit(run)
    print(f"{name:8s} N={N} K={K} {mode}: PLAN {t:6.1f} us {N * K * 2 / t / 1e6:5.2f} TB/s relerr={check():.1e} splits={lib.mi355_linear_packed_splits(M, N, K)}", flush=True)
    for waves in (4, 6, 7, 8):
        row = [f"   waves={waves}:"]
        for fs in (1, 2, 3, 4, 5, 6, 7, 8, 9, 14):
            if fs > max(1, K // 512): continue
            lib.mi355_debug_set(4, waves); lib.mi355_debug_set(2, fs)
            if mode != "silu" and lib.mi355_linear_packed_splits(M, N, K) != fs: continue
            if mode == "silu" and fs > 1: continue
            d = check()
            t = timeit(run)
            row.append(f"s{fs} {t:5.1f}us{'' if d < 2e-2 else ' BAD d=%.3g' % d} |")
        print(" ".join(row), flush=True)
    lib.mi355_debug_set(2, 0); lib.mi355_debug_set(4, 0)
    del w, wp
