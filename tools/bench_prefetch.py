"""Does warming L2 with the next GEMM's weights inside the row-norm launch pay?  Chain of
(rmsnorm -> packed GEMM) pairs over rotating weights, with and without the warm-up riders."""
import sys, torch
sys.path.insert(0, '.')
from sglang_custom_amd import _lib, ops
_lib.use_dev_library()   # ablation switches and probes live in the development build only
lib = _lib.load()
dev = 'cuda'
st = lambda: torch.cuda.current_stream().cuda_stream
# XCD map of consecutive launches
for grids in ((256, 256), (320, 256), (257, 256), (64, 256)):
    outs = [torch.full((g,), -1, dtype=torch.int32, device=dev) for g in grids]
    for o, g in zip(outs, grids): lib.mi355_debug_xcc_map(o.data_ptr(), g, st())
    torch.cuda.synchronize()
    print("grids", grids, [o[:12].tolist() for o in outs], flush=True)

def graph_time(fn, n, reps=3):
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

M, H = 64, 4096
x = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
res = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
g = torch.ones(H, device=dev, dtype=torch.bfloat16)
for name, N, K, act in (("qkv", 6144, 4096, False), ("o", 4096, 4096, False), ("gate_up", 28672, 4096, True), ("down", 4096, 14336, False)):
    nw = max(4, int(1.5e9 // (N * K * 2)))
    ws = [ops.pack_weight_gate_up(torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02) if act
          else ops.pack_weight(torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02) for _ in range(nw)]
    xin = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    def gemm(i):
        return ops.linear_packed_silu_mul(xin, ws[i % nw], N) if act else ops.linear_packed(xin, ws[i % nw], N)
    base_ref = gemm(0).clone()
    for budget_mb in (0, 8, 16, 24, 32):
        def pair(i):
            if budget_mb:
                lib.mi355_prefetch_next(ws[i % nw].data_ptr(), M, N, K, 2, budget_mb << 20)
            ops.fused_add_rmsnorm(x, res, g, 1e-5)
            gemm(i)
        t = graph_time(pair, 40)
        print(f"{name:8s} budget {budget_mb:2d} MB: {t:6.2f} us per (rmsnorm + gemm)", flush=True)
    assert torch.equal(gemm(0), base_ref)
    del ws
