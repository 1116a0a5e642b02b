import sys, time, torch
sys.path.insert(0, '.')
from sglang_custom_amd import ops
from sglang_custom_amd.layers.quantization import w8a8_fp8 as m
x = torch.randn(4096, 4096, device='cuda').bfloat16(); w = torch.randn(6144, 4096, device='cuda').bfloat16() * 0.02
wq, ws = m.quantize_weight_per_channel(w); xq, xs = ops.per_token_quant_fp8(x)
y = m._library_scaled_mm(xq, wq, xs, ws, torch.bfloat16, None)
print("scaled_mm ok:", m._SCALED_MM_OK)
ref = (xq.float() @ wq.float().t()) * xs * ws.t()
if y is not None: print("max err vs fp32 ref", (y.float() - ref).abs().max().item(), "rel", ((y.float() - ref).abs().max() / ref.abs().max()).item())
y2 = ops.fp8_scaled_mm(xq[:64], wq, xs[:64], ws, torch.bfloat16, None)
print("skinny err", (y2.float() - ref[:64]).abs().max().item())
for M in (64,):
    xs_, xq_ = xs[:M].contiguous(), xq[:M].contiguous()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(50): ops.fp8_scaled_mm(xq_, wq, xs_, ws, torch.bfloat16, None)
    torch.cuda.synchronize(); print(f"fp8 skinny qkv M={M}: {(time.time() - t) / 50 * 1e6:.1f} us (host-timed, eager)")
