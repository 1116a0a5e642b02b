"""Prefill projection GEMM (ops.linear_tiled, 256 x 256 tile kernel) against hipBLASLt (F.linear):
correctness on sampled outputs in float64, then interleaved timing rounds on random data."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from sglang_custom_amd import ops

SHAPES_8B = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]


def check(m, n, k, silu, dt=torch.bfloat16, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.rand((m, k), device="cuda", generator=g) * 2 - 1).to(dt)
    w = ((torch.rand((n, k), device="cuda", generator=g) * 2 - 1) * k ** -0.5).to(dt)
    y = ops.linear_tiled(x, w, None, silu_mul=silu)
    torch.cuda.synchronize()
    rows = torch.randint(0, m, (64,), device="cuda", generator=g)
    rows[0], rows[1] = 0, m - 1
    ref = x[rows].double() @ w.double().t()
    if silu:
        gate, up = ref[:, : n // 2].to(dt).float(), ref[:, n // 2:].to(dt).float()
        ref = (F.silu(gate).to(dt).float() * up).double()
    err = (y[rows].double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    return err, scale


def check_fp8(m, n, k, silu, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.rand((m, k), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand((n, k), device="cuda", generator=g) * 2 - 1) * k ** -0.5).to(torch.bfloat16)
    xq, xs = ops.per_token_quant_fp8(x)
    ws = (w.float().abs().amax(1, keepdim=True).clamp(min=1e-10) / 448.0)
    wq = (w.float() / ws).clamp(-448, 448).to(torch.float8_e4m3fn)
    y = ops.fp8_scaled_mm_tiled(xq, wq, xs, ws.contiguous(), torch.bfloat16, None, silu_mul=silu)
    torch.cuda.synchronize()
    rows = torch.randint(0, m, (64,), device="cuda", generator=g)
    rows[0], rows[1] = 0, m - 1
    ref = (xq[rows].double() @ wq.double().t()) * xs[rows].double().view(-1, 1) * ws.double().view(1, -1)
    if silu:
        gate, up = ref[:, : n // 2].to(torch.bfloat16).float(), ref[:, n // 2:].to(torch.bfloat16).float()
        ref = (F.silu(gate).to(torch.bfloat16).float() * up).double()
    return (y[rows].double() - ref).abs().max().item(), ref.abs().max().item()


def timeit(fn, rounds):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(rounds):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / rounds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=65536)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="time the fp8 W8A8 kernel against torch._scaled_mm")
    ap.add_argument("--lib", default=None, help="another build of the library (development A/B)")
    a = ap.parse_args()
    if a.lib:
        from sglang_custom_amd import _lib
        _lib.use_dev_library(os.path.abspath(a.lib))
    for (m, n, k, silu) in [(256, 256, 128, False), (300, 272, 192, False), (1000, 512, 1024, True),
                            (4096, 6144, 4096, False), (2048, 28672, 4096, True), (777, 4096, 14336, False)][:0 if a.no_check else None]:
        err, scale = check(m, n, k, silu)
        print(f"check m={m} n={n} k={k} silu={silu}: max err {err:.3e} (|ref| max {scale:.3f})", flush=True)
        assert err < 0.02 * max(scale, 1.0), "mismatch"
    if not a.no_check:
        for (m, n, k, silu) in [(256, 256, 256, False), (300, 272, 384, False), (1000, 512, 1024, True),
                                (4096, 6144, 4096, False), (777, 4096, 14336, False)]:
            err, scale = check_fp8(m, n, k, silu)
            print(f"check fp8 m={m} n={n} k={k} silu={silu}: max err {err:.3e} (|ref| max {scale:.3f})", flush=True)
            assert err < 0.02 * max(scale, 1.0), "mismatch"
    if a.check_only:
        return
    m = a.m
    if a.fp8:
        for name, n, k in SHAPES_8B:
            x = (torch.rand((m, k), device="cuda") * 2 - 1).to(torch.bfloat16)
            w = ((torch.rand((n, k), device="cuda") * 2 - 1) * k ** -0.5).to(torch.bfloat16)
            xq, xs = ops.per_token_quant_fp8(x)
            ws = (w.float().abs().amax(1, keepdim=True).clamp(min=1e-10) / 448.0).contiguous()
            wq = (w.float() / ws).clamp(-448, 448).to(torch.float8_e4m3fn)
            del x, w
            fl = 2.0 * m * n * k
            silu = name == "gate_up"
            ours = lambda: ops.fp8_scaled_mm_tiled(xq, wq, xs, ws, torch.bfloat16, None, silu_mul=silu)
            mm = lambda: torch._scaled_mm(xq, wq.t(), scale_a=xs, scale_b=ws.t(), out_dtype=torch.bfloat16)
            lib = (lambda: ops.silu_and_mul(mm())) if silu else mm
            for f in (ours, lib):
                f()
            torch.cuda.synchronize()
            to, tl = [], []
            for _ in range(a.rounds):
                to.append(timeit(ours, a.reps))
                tl.append(timeit(lib, a.reps))
            mo, ml = sorted(to)[len(to) // 2], sorted(tl)[len(tl) // 2]
            print(f"fp8 {name:8s} M={m} N={n} K={k}{' +silu_mul' if silu else ''}: tiled {mo:.3f} ms = "
                  f"{fl / mo / 1e9:.0f} TFLOP/s | _scaled_mm{'+silu_and_mul' if silu else ''} {ml:.3f} ms = "
                  f"{fl / ml / 1e9:.0f} TFLOP/s | ratio {ml / mo:.3f}", flush=True)
        return
    for name, n, k in SHAPES_8B:
        x = (torch.rand((m, k), device="cuda") * 2 - 1).to(torch.bfloat16)
        w = ((torch.rand((n, k), device="cuda") * 2 - 1) * k ** -0.5).to(torch.bfloat16)
        fl = 2.0 * m * n * k
        silu = name == "gate_up"
        ours = lambda: ops.linear_tiled(x, w, None, silu_mul=silu)
        lib = (lambda: ops.silu_and_mul(F.linear(x, w))) if silu else (lambda: F.linear(x, w))
        for f in (ours, lib):
            f()
        torch.cuda.synchronize()
        to, tl = [], []
        for _ in range(a.rounds):
            to.append(timeit(ours, a.reps))
            tl.append(timeit(lib, a.reps))
        bo, bl = min(to), min(tl)
        mo, ml = sorted(to)[len(to) // 2], sorted(tl)[len(tl) // 2]
        print(f"{name:8s} M={m} N={n} K={k}{' +silu_mul' if silu else ''}: tiled {mo:.3f} ms (best {bo:.3f}) = "
              f"{fl / mo / 1e9:.0f} TFLOP/s | hipBLASLt{'+silu_and_mul' if silu else ''} {ml:.3f} ms (best {bl:.3f}) = "
              f"{fl / ml / 1e9:.0f} TFLOP/s | ratio {ml / mo:.3f}", flush=True)


if __name__ == "__main__":
    main()
