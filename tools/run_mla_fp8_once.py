"""One fp8-latent-cache MLA decode launch per call at the DeepSeek-V3 TP=8 shard shape (64 x 16 heads x 4096 keys, 8 splits,
in-launch merge), a few calls: the program rocprofv3 counter passes are taken over (tools: profiles/r03_mla_fp8_ring_pmc.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops
bs, H, ctx, rank, rope, ns = 64, 16, 4096, 512, 64, 8
D = rank + rope
torch.manual_seed(0)
kv8 = (torch.randn(bs * ctx + 16, 1, D, device='cuda') * 2).to(torch.float8_e4m3fn)
q = torch.randn(bs, H, D, device='cuda').bfloat16()
o = torch.empty(bs, H, rank, device='cuda', dtype=torch.bfloat16)
r2t = (1 + torch.arange(bs * ctx, device='cuda', dtype=torch.int32)).view(bs, ctx).contiguous()
rpi = torch.arange(bs, dtype=torch.int64, device='cuda')
sl = torch.full((bs,), ctx, dtype=torch.int64, device='cuda')
lg = torch.empty(bs * H * ns * rank, dtype=torch.float32, device='cuda'); ls = torch.empty(bs * H * ns, dtype=torch.float32, device='cuda')
tk = torch.zeros(bs * H, dtype=torch.int32, device='cuda')
for _ in range(5):
    ops.decode_attention_mla(q, kv8, o, r2t, rpi, sl, lg, ls, ns, rank, D ** -0.5, tickets=tk)
torch.cuda.synchronize()
print("done", float(o.float().abs().mean()))
