"""Dev tool: MLA decode (64 sequences x 16 heads, 8 KV splits, in-launch merge) with ONE KV split per workgroup against TWO
(two teams of four waves whose results meet in LDS; development switch 10), over a 16-bit and an fp8 latent cache, by context."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops, _lib
_lib.use_dev_library(); lib = _lib.load()
lib.mi355_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
bs, H, rank, rope = 64, 16, 512, 64
D = rank + rope
def timeit(fn, reps=8):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3
for ctx in [int(a) for a in sys.argv[1:]] or (64, 1024, 4096, 8192):
    kv = [torch.randn(bs * ctx + 16, 1, D, device='cuda').bfloat16() for _ in range(3)]
    kv8 = [(t.float() * 2).to(torch.float8_e4m3fn) for t in kv]
    q = torch.randn(bs, H, D, device='cuda').bfloat16()
    o = torch.empty(bs, H, rank, device='cuda', dtype=torch.bfloat16)
    r2t = (1 + torch.arange(bs * ctx, device='cuda', dtype=torch.int32)).view(bs, ctx).contiguous()
    rpi = torch.arange(bs, dtype=torch.int64, device='cuda'); sl = torch.full((bs,), ctx, dtype=torch.int64, device='cuda')
    ns = 8
    lg = torch.empty(bs * H * ns * rank, dtype=torch.float32, device='cuda'); ls = torch.empty(bs * H * ns, dtype=torch.float32, device='cuda')
    tk = torch.zeros(bs * H, dtype=torch.int32, device='cuda')
    row = [f"ctx={ctx}:"]
    for name, cache in (("16-bit", kv), ("fp8", kv8)):
        for teams in (1, 0):
            lib.mi355_debug_set(10, teams)
            t = timeit(lambda i: ops.decode_attention_mla(q, cache[i % 3], o, r2t, rpi, sl, lg, ls, ns, rank, D ** -0.5, tickets=tk))
            row.append(f"{name} {'one team ' if teams else 'two teams'} {t:6.1f} us |")
    print(" ".join(row), flush=True)
    del kv, kv8
