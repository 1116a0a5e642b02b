"""Dev tool: prefill (extend) attention over a CACHED PREFIX (chunked prefill, prefix-cache hits): bs sequences, each
with `pre` cached tokens read through the page table and `ext` new tokens; against the same total work without a
prefix.  usage: bench_extend_prefix.py [bs pre ext hq hkv]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops
a = [int(x) for x in sys.argv[1:]]
bs, pre, ext, hq, hkv = a if len(a) == 5 else (8, 2048, 2048, 32, 8)
D = 128
torch.manual_seed(0)
T = bs * ext
q = torch.randn(T, hq, D, device='cuda').bfloat16()
k = torch.randn(T, hkv, D, device='cuda').bfloat16()
v = torch.randn(T, hkv, D, device='cuda').bfloat16()
o = torch.empty_like(q)
slots = bs * pre + 16
kb = torch.randn(slots, hkv, D, device='cuda').bfloat16(); vb = torch.randn(slots, hkv, D, device='cuda').bfloat16()
qo = torch.arange(0, T + 1, ext, dtype=torch.int32, device='cuda')
def run(order):
    kvp = torch.arange(0, bs * pre + 1, pre, dtype=torch.int32, device='cuda') if pre else torch.zeros(bs + 1, dtype=torch.int32, device='cuda')
    kvi = order.to(torch.int32) if pre else torch.zeros(1, dtype=torch.int32, device='cuda')
    def f():
        ops.extend_attention_fwd(q, k, v, o, kb, vb, qo, kvp, kvi, None, True, None, ext, D ** -0.5)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
flops = 4 * bs * hq * D * (ext * pre + ext * ext / 2)
if pre:
    t = run(1 + torch.arange(bs * pre, device='cuda'))
    print(f"bs={bs} prefix={pre} extend={ext} {hq}q/{hkv}kv, prefix slots in order: {t:.3f} ms  {flops / t / 1e9:.0f} TFLOP/s", flush=True)
    t = run(1 + torch.randperm(bs * pre, device='cuda'))
    print(f"bs={bs} prefix={pre} extend={ext} {hq}q/{hkv}kv, prefix slots shuffled : {t:.3f} ms  {flops / t / 1e9:.0f} TFLOP/s", flush=True)
    # check one sequence against SDPA
    import torch.nn.functional as F
    order = 1 + torch.arange(bs * pre, device='cuda')
    run(order)
    b = bs - 1
    kk = torch.cat([kb[order[b * pre:(b + 1) * pre]], k[b * ext:(b + 1) * ext]]).transpose(0, 1).float()
    vv = torch.cat([vb[order[b * pre:(b + 1) * pre]], v[b * ext:(b + 1) * ext]]).transpose(0, 1).float()
    qq = q[b * ext:(b + 1) * ext].transpose(0, 1).float()
    mask = torch.ones(ext, pre + ext, dtype=torch.bool, device='cuda').tril(pre)
    ref = F.scaled_dot_product_attention(qq[None], kk[None], vv[None], attn_mask=mask, enable_gqa=True)[0].transpose(0, 1)
    print(f"   max err vs SDPA fp32 = {(o[b * ext:(b + 1) * ext].float() - ref).abs().max().item():.4f}", flush=True)
else:
    t = run(None)
    print(f"bs={bs} no prefix, extend={ext} {hq}q/{hkv}kv: {t:.3f} ms  {flops / t / 1e9:.0f} TFLOP/s", flush=True)
