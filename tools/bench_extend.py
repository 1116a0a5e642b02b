"""Dev tool: prefill (extend) attention at the bench shape: 64 x 1024 tokens, 32q/8kv x 128, causal."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
from sglang_custom_amd import ops, _lib
_lib.use_dev_library(os.environ.get('MI355_AB_LIB'))   # MI355_AB_LIB: a development build of another source tree; ablation switches live in the development build only
lib = _lib.load()
# usage: bench_extend.py [bs L [hq hkv Dq Dv]]   (MLA prefill in its multi-head form: 16 16 192 128)
_a = [int(x) for x in sys.argv[1:]]
bs, L = (_a + [64, 1024])[:2] if len(_a) < 2 else _a[:2]
hq, hkv, D, Dv = _a[2:6] if len(_a) >= 6 else (32, 8, 128, 128)
torch.manual_seed(0)
T = bs * L
q = torch.randn(T, hq, D, device='cuda').bfloat16()
k = torch.randn(T, hkv, D, device='cuda').bfloat16()
v = torch.randn(T, hkv, Dv, device='cuda').bfloat16()
o = torch.empty(T, hq, Dv, device='cuda', dtype=torch.bfloat16)
kb = torch.zeros(16, hkv, D, device='cuda').bfloat16(); vb = torch.zeros(16, hkv, Dv, device='cuda').bfloat16()
qo = torch.arange(0, T + 1, L, dtype=torch.int32, device='cuda')
kvi = torch.zeros(bs + 1, dtype=torch.int32, device='cuda'); kidx = torch.zeros(1, dtype=torch.int32, device='cuda')
def run():
    ops.extend_attention_fwd(q, k, v, o, kb, vb, qo, kvi, kidx, None, True, None, L, D ** -0.5)
import torch.nn.functional as F
for variant, name in ((1, "16x16x32 tile, 64 rows per workgroup (ablation)"), (0, "32x32x16 tile, 128 rows per workgroup (product)")):
    if variant == 1 and D != 128: continue
    lib.mi355_debug_set(8, variant)
    for _ in range(30): run()   # clocks settle: a cold GPU times the same kernel 20-30 % slower
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30
    flops = 2 * bs * hq * (D + Dv) * L * L / 2
    print(f"{name}: bs={bs} L={L}: {t:.3f} ms  {flops / t / 1e9:.1f} TFLOP/s", flush=True)
    for b in (0, bs - 1):   # a few sequences against SDPA
        qq = q[b * L:(b + 1) * L].transpose(0, 1).float(); kk = k[b * L:(b + 1) * L].transpose(0, 1).float(); vv = v[b * L:(b + 1) * L].transpose(0, 1).float()
        ref = F.scaled_dot_product_attention(qq[None], kk[None], vv[None], is_causal=True, enable_gqa=True)[0].transpose(0, 1)
        err = (o[b * L:(b + 1) * L].float() - ref).abs().max().item()
        print(f"   seq {b}: max err vs SDPA fp32 = {err:.4f}", flush=True)
