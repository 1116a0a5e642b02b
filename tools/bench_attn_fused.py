"""Dev tool: the decode attention launch that starts from the qkv projection's split-K partials (sum, RoPE,
KV-cache write, attention) against the plain attention launch on ready-made q, at the bench shape.
usage: [QKV_SPLITS=4] [KV_SPLITS=1] python tools/bench_attn_fused.py [bs hq hkv [ctx]]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops, _lib
_lib.load()
bs, ctx, hq, hkv, D = 64, 1064, 32, 8, 128
nsplit = int(os.environ.get('QKV_SPLITS', '4'))
if len(sys.argv) > 3: bs, hq, hkv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4: ctx = int(sys.argv[4])
NL = 3
slots = bs * (ctx + 80) + 16
torch.manual_seed(0)
kb = [torch.randn(slots, hkv, D, device='cuda').bfloat16() for _ in range(NL)]
vb = [torch.randn(slots, hkv, D, device='cuda').bfloat16() for _ in range(NL)]
r2t = torch.zeros(bs, ctx + 16, dtype=torch.int32, device='cuda')
for b in range(bs):
    r2t[b, :ctx - 1] = torch.arange(1 + b * (ctx + 40), b * (ctx + 40) + ctx)
    r2t[b, ctx - 1] = 1 + bs * (ctx + 40) + b     # the new token's slot
rpi = torch.arange(bs, dtype=torch.int64, device='cuda')
sl = torch.full((bs,), ctx, dtype=torch.int64, device='cuda')
loc = r2t[:, ctx - 1].to(torch.int64).contiguous()
pos = (sl - 1).contiguous()
row = (hq + 2 * hkv) * D
part = [torch.randn(nsplit, bs, row, device='cuda') * 0.3 for _ in range(NL)]
cache = torch.cat([torch.randn(ctx + 8, D // 2, device='cuda').cos(), torch.randn(ctx + 8, D // 2, device='cuda').sin()], 1).contiguous()
q = torch.randn(bs, hq, D, device='cuda').bfloat16()
o = torch.empty(bs, hq, D, device='cuda', dtype=torch.bfloat16)
def timeit(fn, reps=12):
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
nbytes = bs * ctx * hkv * D * 2 * 2
ks = int(os.environ.get('KV_SPLITS', '1'))       # KV splits, merged inside the launch
lg = torch.empty(bs * hq * ks * D, dtype=torch.float32, device='cuda') if ks > 1 else None
ls = torch.empty(bs * hq * ks, dtype=torch.float32, device='cuda') if ks > 1 else None
tk = torch.zeros(bs * hq, dtype=torch.int32, device='cuda') if ks > 1 else None
t0 = timeit(lambda i: ops.decode_attention_paged(q, kb[i % NL], vb[i % NL], o, r2t, rpi, sl, lg, ls, ks, D ** -0.5, tickets=tk))
t1 = timeit(lambda i: ops.decode_attention_fused_qkv(part[i % NL], pos, loc, cache, kb[i % NL], vb[i % NL], o, r2t, rpi, sl, lg, ls, ks, D ** -0.5, tickets=tk))
print(f"{ks} KV split(s): plain {t0:6.1f} us {nbytes / t0 / 1e6:5.2f} TB/s   from {nsplit} qkv partials {t1:6.1f} us {nbytes / t1 / 1e6:5.2f} TB/s", flush=True)
