import sys, math, torch, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import cases
from sglang_custom_amd import ops, _lib
lib = _lib.load()
lib.mi355_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
for (M, N, K) in [(16, 4100, 1024), (16, 4100, 512), (16, 4160, 1024), (16, 4112, 1024), (64, 4100, 1024), (16, 132, 1024)]:
    x = cases.randn((M, K), 7, torch.bfloat16); w = cases.randn((N, K), 8, torch.bfloat16, 1.0 / math.sqrt(K))
    exact = x.double() @ w.double().t()
    for variant in (0, 1):
        for fs in (0, 1, 2):
            lib.mi355_debug_set(0, variant); lib.mi355_debug_set(2, fs)
            got = ops.linear(x.cuda(), w.cuda()).cpu().double()
            gotp = ops.linear_packed(x.cuda(), ops.pack_weight(w.cuda()), N).cpu().double()
            bad = ((got - exact).abs() > 0.05); badp = ((gotp - exact).abs() > 0.05)
            print((M, N, K), "variant", variant, "splits", fs, "bad cols", bad.any(0).nonzero().flatten().tolist()[:6], "packed bad", badp.any(0).nonzero().flatten().tolist()[:6])
lib.mi355_debug_set(0, 0); lib.mi355_debug_set(2, 0)
