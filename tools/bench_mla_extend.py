"""MLA extend over cached rows in the absorbed form (one latent "kv head" of 512 + 64 features shared by all
query heads): new tokens of each request attend to the request's prefix and, causally, to each other.
Times the per-token multi-query pass (decode kernel, one launch item per new token) and, when the library
has it, the tiled kernel.  Usage: python tools/bench_mla_extend.py [n_seq n_new prefix heads]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import _lib, ops  # noqa: E402

_lib.load()
cases = [(1, 4096, 0, 16), (1, 4096, 4096, 16), (4, 1024, 1024, 16), (1, 1024, 0, 128), (16, 256, 3840, 16)]
if len(sys.argv) > 4:
    cases = [tuple(int(a) for a in sys.argv[1:5])]
rank, rope = 512, 64
D = rank + rope
dt = torch.bfloat16


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for (B, n_new, prefix, H) in cases:
    ctx = prefix + n_new
    torch.manual_seed(0)
    kv = torch.randn(B * ctx + 1, 1, D, device="cuda").to(dt)
    q = torch.randn(B * n_new, H, D, device="cuda").to(dt)
    r2t = (1 + torch.arange(B * ctx, device="cuda", dtype=torch.int32)).view(B, ctx).contiguous()
    rpi = torch.arange(B, dtype=torch.int64, device="cuda")
    tok_req = rpi.repeat_interleave(n_new)
    tok_len = (prefix + 1 + torch.arange(n_new, device="cuda", dtype=torch.int64)).repeat(B)
    o = torch.empty(B * n_new, H, rank, device="cuda", dtype=dt)
    flops = 2.0 * H * (D + rank) * float(tok_len.sum().item())

    def per_token():
        ops.decode_attention_mla(q, kv, o, r2t, tok_req, tok_len, None, None, 1, rank, D ** -0.5)

    t0 = timeit(per_token)
    line = f"B={B} new={n_new} prefix={prefix} H={H}: per-token {t0:9.1f} us ({flops / t0 / 1e6:6.1f} TFLOP/s)"
    if hasattr(ops, "extend_attention_mla"):
        o2 = torch.empty_like(o)
        sl = torch.full((B,), ctx, dtype=torch.int64, device="cuda")
        qo = (torch.arange(B + 1, device="cuda") * n_new).to(torch.int32)

        def tiled():
            ops.extend_attention_mla(q, kv, o2, r2t, rpi, sl, qo, n_new, rank, D ** -0.5)

        t1 = timeit(tiled)
        err = (o.float() - o2.float()).abs().max().item()
        line += f"  tiled {t1:9.1f} us ({flops / t1 / 1e6:6.1f} TFLOP/s)  x{t0 / t1:.2f}  max|diff| {err:.4f}"
    print(line, flush=True)
