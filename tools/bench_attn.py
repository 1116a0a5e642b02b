"""Dev tool: decode attention micro-benchmark at the bench shape (bs=64, ctx~1064, 32q/8kv x128)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops, _lib
_lib.use_dev_library(os.environ.get('MI355_AB_LIB'))   # MI355_AB_LIB: a development build of another source tree; ablation switches and probes live in the development build only
lib = _lib.load()
lib.mi355_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
bs, ctx, hq, hkv, D = 64, 1064, 32, 8, 128
if len(sys.argv) > 3: bs, hq, hkv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4: ctx = int(sys.argv[4])
SP = int(sys.argv[5]) if len(sys.argv) > 5 else 1024   # slot spacing between sequences' prompt regions
NL = 3
SP = max(SP, ctx)
slots = int(os.environ.get('SLOTS', bs * max(1100, SP + 80) + 16))   # SLOTS: pool rows (the head stride of the head-major layout)
torch.manual_seed(0)
kb = [torch.randn(slots, hkv, D, device='cuda').bfloat16() for _ in range(NL)]
vb = [torch.randn(slots, hkv, D, device='cuda').bfloat16() for _ in range(NL)]
import os
if os.environ.get('KV8'):
    kb = [t.to(torch.float8_e4m3fn) for t in kb]; vb = [t.to(torch.float8_e4m3fn) for t in vb]
if os.environ.get('LAYOUT', 'hm') == 'hm':   # the pool's layout on the GPU: [heads, slots, 2, d], keys | values (LAYOUT=tm: the reference's)
    def _hm(k_, v_):
        kv = torch.empty((hkv, slots, 2, D), dtype=k_.dtype, device='cuda')
        k2, v2 = kv[:, :, 0, :].permute(1, 0, 2), kv[:, :, 1, :].permute(1, 0, 2)
        k2.copy_(k_); v2.copy_(v_)
        return k2, v2
    pairs = [_hm(k_, v_) for k_, v_ in zip(kb, vb)]
    kb, vb = [p_[0] for p_ in pairs], [p_[1] for p_ in pairs]
q = torch.randn(bs, hq, D, device='cuda').bfloat16()
o = torch.empty_like(q)
# prompt tokens contiguous per request, decode tokens interleaved across requests (as the allocator hands them out)
r2t = torch.zeros(bs, max(1200, ctx + 16), dtype=torch.int32, device='cuda')
for b in range(bs):
    PL = min(ctx, max(1024, ctx - 40))   # prompt length: the last few tokens are decode tokens
    r2t[b, :PL] = torch.arange(1 + b * SP, 1 + b * SP + PL)
    if ctx > PL: r2t[b, PL:ctx] = 1 + bs * SP + torch.arange(ctx - PL, device='cuda') * bs + b
assert int(r2t.max()) < slots, f"SLOTS={slots} is smaller than the highest slot in use ({int(r2t.max())})"   # checked on the host: never index past the pool
rpi = torch.arange(bs, dtype=torch.int64, device='cuda')
sl = torch.full((bs,), ctx, dtype=torch.int64, device='cuda')
SPLIT_LIST = [int(x) for x in os.environ.get('SPLITS', '1,2,3,4').split(',')]
MAXS = max(SPLIT_LIST)   # the partials of the widest split plan must fit
logits = torch.empty(bs, hq, MAXS, D, dtype=torch.float32, device='cuda'); lse = torch.empty(bs, hq, MAXS, dtype=torch.float32, device='cuda')
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
# reference
idx = r2t[:, :ctx].long()
K = kb[0][idx].float(); V = vb[0][idx].float()            # [bs, ctx, hkv, D]
qg = q.float().view(bs, hkv, hq // hkv, D)
s = torch.einsum('bhgd,bkhd->bhgk', qg, K) * D ** -0.5
ref = torch.einsum('bhgk,bkhd->bhgd', s.softmax(-1), V).reshape(bs, hq, D)
nbytes = bs * ctx * hkv * D * 2 * kb[0].element_size()
# TICKETS=1: the KV splits are merged inside the attention launch (arrival counters) instead of by a second launch
tk = torch.zeros(bs * hq, dtype=torch.int32, device='cuda') if os.environ.get('TICKETS') else None
for dbg in ((0,) if os.environ.get('ONLY0') else (3, 0)):
    lib.mi355_debug_set(6, dbg)   # 3 = K rows straight into fragments (ablation), 0 = whole K rows staged through LDS (product)
    for ns in SPLIT_LIST:
        o.zero_()
        ops.decode_attention_paged(q, kb[0], vb[0], o, r2t, rpi, sl, logits, lse, ns, D ** -0.5, tickets=tk)
        err = (o.float() - ref).abs().max().item()
        t = timeit(lambda i: ops.decode_attention_paged(q, kb[i % NL], vb[i % NL], o, r2t, rpi, sl, logits, lse, ns, D ** -0.5, tickets=tk))
        left = int(tk.abs().sum().item()) if tk is not None else 0
        print(f"dbg={dbg} splits={ns}{' in-kernel merge' if tk is not None else ''}: {t:6.1f} us  {nbytes / t / 1e6:5.2f} TB/s  err={err:.4f}"
              + (f"  tickets left {left}" if left else ""), flush=True)
