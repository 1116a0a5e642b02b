"""MLA decode attention micro-benchmark (DeepSeek-V3 TP=8 shard: 16 heads, latent 512 + 64 rope)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops, _lib
_lib.use_dev_library(os.environ.get('MI355_AB_LIB'))   # MI355_AB_LIB: a development build of another source tree; ablation switches and probes live in the development build only
lib = _lib.load()
bs, H, ctx = 64, 16, 4096
if len(sys.argv) > 3: bs, H, ctx = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rank, rope = 512, 64
D = rank + rope
NL = 3
slots = bs * ctx + 16
torch.manual_seed(0)
kv = [torch.randn(slots, 1, D, device='cuda').bfloat16() for _ in range(NL)]
q = torch.randn(bs, H, D, device='cuda').bfloat16()
o = torch.empty(bs, H, rank, device='cuda', dtype=torch.bfloat16)
r2t = (1 + torch.arange(bs * ctx, device='cuda', dtype=torch.int32)).view(bs, ctx).contiguous()
rpi = torch.arange(bs, dtype=torch.int64, device='cuda')
sl = torch.full((bs,), ctx, dtype=torch.int64, device='cuda')
SPLIT_LIST = [int(x) for x in os.environ.get('SPLITS', '1,2,4,8').split(',')]
MAXS = max(SPLIT_LIST)   # the partials of the widest split plan must fit
lg = torch.empty(bs * H * MAXS * rank, dtype=torch.float32, device='cuda'); ls = torch.empty(bs * H * MAXS, dtype=torch.float32, device='cuda')
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
nb = min(bs, 4)
K = kv[0][r2t[:nb].long(), 0].float()
s = torch.einsum('bhd,bkd->bhk', q[:nb].float(), K) * D ** -0.5
ref = torch.einsum('bhk,bkd->bhd', s.softmax(-1), K[..., :rank])
nbytes = bs * ctx * D * 2
# TICKETS=1: the KV splits are merged inside the attention launch instead of by a second launch
tk = torch.zeros(bs * H, dtype=torch.int32, device='cuda') if os.environ.get('TICKETS') else None
for dbg in ((0,) if os.environ.get("MLA_AB") else (1, 2, 5, 0)):
    lib.mi355_debug_set(6, dbg)   # 1 = generic wave-per-head kernel, 2 = MFMA kernel with rows in registers, 0 = rows through the LDS ring (product)
    for ns in SPLIT_LIST:
        if dbg == 1 and ns not in (1, 4): continue
        if len(sys.argv) > 4 and ns != int(sys.argv[4]): continue
        o.zero_()
        ops.decode_attention_mla(q, kv[0], o, r2t, rpi, sl, lg, ls, ns, rank, D ** -0.5, tickets=tk)
        err = (o[:nb].float() - ref).abs().max().item()
        t = timeit(lambda i: ops.decode_attention_mla(q, kv[i % NL], o, r2t, rpi, sl, lg, ls, ns, rank, D ** -0.5, tickets=tk))
        print(f"variant={dbg} splits={ns}{' in-kernel merge' if tk is not None and dbg in (0, 5) else ''}: {t:7.1f} us  {nbytes / t / 1e6:5.2f} TB/s  err={err:.4f}", flush=True)
# e4m3fn latent cache (kv_cache_dtype fp8_e4m3): half the row bytes
kv8 = [(t.float() * 2).to(torch.float8_e4m3fn) for t in kv]
K8 = kv8[0][r2t[:nb].long(), 0].float()
s8 = torch.einsum('bhd,bkd->bhk', q[:nb].float(), K8) * D ** -0.5
ref8 = torch.einsum('bhk,bkd->bhd', s8.softmax(-1), K8[..., :rank])
for dbg in (2, 0):   # 2 = register kernel (widening on use, merge launch), 0 = fp8 rows through the LDS ring (product)
    lib.mi355_debug_set(6, dbg)
    for ns in SPLIT_LIST:
        o.zero_()
        tk8 = tk if dbg == 0 else None
        ops.decode_attention_mla(q, kv8[0], o, r2t, rpi, sl, lg, ls, ns, rank, D ** -0.5, tickets=tk8)
        err = (o[:nb].float() - ref8).abs().max().item()
        t = timeit(lambda i: ops.decode_attention_mla(q, kv8[i % NL], o, r2t, rpi, sl, lg, ls, ns, rank, D ** -0.5, tickets=tk8))
        print(f"fp8 latent cache variant={dbg} splits={ns}{' in-kernel merge' if tk8 is not None else ''}: {t:7.1f} us  {nbytes / 2 / t / 1e6:5.2f} TB/s of fp8 rows  err={err:.4f}", flush=True)
lib.mi355_debug_set(6, 0)
# RoPE + latent-row write inside the launch against the three launches it replaces (rope, cache write, attention)
lib.mi355_debug_set(6, 0)
pos_t = sl - 1
loc = r2t[rpi, (sl - 1)].to(torch.int64)
cs = torch.cat([torch.randn(ctx + 8, rope // 2, device='cuda').cos(), torch.randn(ctx + 8, rope // 2, device='cuda').sin()], 1).contiguous()
k_nope = torch.randn(bs, rank, device='cuda').bfloat16()
lat = torch.randn(bs, 1536 + D, device='cuda').bfloat16()
k_pe_v = lat[:, 1536 + rank:]
for ns in SPLIT_LIST:
    if ns not in (4, 8): continue
    def three(i):
        q_pe = q[..., rank:].contiguous().view(bs, -1)
        k_pe = k_pe_v.contiguous()
        ops.rope_inplace(pos_t, q_pe, k_pe, rope, cs, False)
        q[..., rank:] = q_pe.view(bs, H, rope)
        row = torch.cat([k_nope, k_pe], -1).view(bs, 1, D)
        ops.set_kv_buffer(kv[i % NL], kv[i % NL], loc, row, row)
        ops.decode_attention_mla(q, kv[i % NL], o, r2t, rpi, sl, lg, ls, ns, rank, D ** -0.5, tickets=tk)
    def one(i):
        ops.decode_attention_mla_fused_rope(q, k_nope, k_pe_v, pos_t, loc, cs, False, kv[i % NL], o, r2t, rpi, sl, lg, ls, ns, rank,
                                            D ** -0.5, tickets=tk)
    t3, t1 = timeit(three), timeit(one)
    print(f"splits={ns}: rope + cache write + attention as the model issues them {t3:7.1f} us; one launch {t1:7.1f} us", flush=True)
