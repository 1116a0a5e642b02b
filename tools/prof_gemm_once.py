"""One launch each of the tiled prefill GEMM and hipBLASLt on the same operands (for rocprofv3 passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from sglang_custom_amd import ops

m, n, k = 65536, 4096, 14336
x = (torch.rand((m, k), device="cuda") * 2 - 1).to(torch.bfloat16)
w = ((torch.rand((n, k), device="cuda") * 2 - 1) * k ** -0.5).to(torch.bfloat16)
for _ in range(6):
    ops.linear_tiled(x, w)
    F.linear(x, w)
torch.cuda.synchronize()
