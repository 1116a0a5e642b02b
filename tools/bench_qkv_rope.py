"""qkv projection + RoPE + KV-cache write at prefill sizes: the fused launch (mi355_qkv_proj_rope_tiled)
against the projection followed by mi355_rope_inplace.  Usage: python tools/bench_qkv_rope.py [--fp8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp8", action="store_true")
    args = ap.parse_args()
    _lib.load()
    dt = torch.bfloat16
    for (m, k, hq, hkv) in [(65536, 4096, 32, 8), (16384, 4096, 32, 8), (65536, 8192, 8, 1)]:
        n = (hq + 2 * hkv) * 128
        x = torch.randn(m, k, device="cuda").to(dt)
        w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(dt)
        inv = 1.0 / (500000.0 ** (torch.arange(0, 128, 2, device="cuda").float() / 128))
        fr = torch.arange(8192, device="cuda").float()[:, None] * inv[None]
        cs = torch.cat([fr.cos(), fr.sin()], dim=-1).contiguous()
        pos = (torch.arange(m, device="cuda") % 1024).long()
        loc = torch.arange(m, device="cuda").long()
        kb = torch.zeros(m, hkv, 128, device="cuda", dtype=dt)
        vb = torch.zeros_like(kb)
        if args.fp8:
            from sglang_custom_amd.layers.quantization.w8a8_fp8 import quantize_weight_per_channel
            xq, xs = ops.per_token_quant_fp8(x)
            wq, ws = quantize_weight_per_channel(w)
            ws = ws.contiguous().view(-1)

            def unfused():
                y = ops.fp8_scaled_mm_tiled(xq, wq, xs, ws, dt)
                q, kk, v = y.split([hq * 128, hkv * 128, hkv * 128], dim=-1)
                ops.rope_inplace(pos, q, kk, 128, cs, True, v=v, k_buffer=kb, v_buffer=vb, loc=loc)
                return y

            def fused():
                return ops.qkv_proj_rope_tiled(xq, wq, hq, hkv, 128, pos, cs, True, kb, vb, loc, x_scale=xs,
                                               w_scale=ws, out_dtype=dt)

            def gemm():
                return ops.fp8_scaled_mm_tiled(xq, wq, xs, ws, dt)
        else:
            def unfused():
                y = ops.linear_tiled(x, w)
                q, kk, v = y.split([hq * 128, hkv * 128, hkv * 128], dim=-1)
                ops.rope_inplace(pos, q, kk, 128, cs, True, v=v, k_buffer=kb, v_buffer=vb, loc=loc)
                return y

            def fused():
                return ops.qkv_proj_rope_tiled(x, w, hq, hkv, 128, pos, cs, True, kb, vb, loc)

            def gemm():
                return ops.linear_tiled(x, w)
        assert torch.equal(unfused(), fused())
        tg, tu, tf = timeit(gemm), timeit(unfused), timeit(fused)
        fl = 2.0 * m * n * k
        print(f"{'fp8' if args.fp8 else 'bf16'} M={m} N={n} K={k}: gemm {tg:8.1f} us ({fl / tg / 1e6:7.1f} TFLOP/s)  "
              f"gemm+rope {tu:8.1f} us  fused {tf:8.1f} us ({fl / tf / 1e6:7.1f} TFLOP/s)  fused/unfused {tf / tu:.3f}",
              flush=True)


if __name__ == "__main__":
    main()
