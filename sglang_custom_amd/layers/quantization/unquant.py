"""UnquantizedLinearMethod.apply(layer, x, bias) = F.linear
(python/sglang/srt/layers/quantization/unquant.py:110-122).

Decode-sized inputs (M <= 64 rows) and every fp32 input go to the hand-written kernels behind
mi355_linear (weight-streaming MFMA / exact-fp32 tiled).  Large-M bf16/f16 GEMMs (prefill) are
plain library GEMMs: torch.nn.functional.linear -> hipBLASLt."""
from typing import Optional

import torch
import torch.nn.functional as F

from ...ops_backend import get_ops

SKINNY_MAX_M = 64


class UnquantizedLinearMethod:
    def process_weights_after_loading(self, layer) -> None:
        pass

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        ops = get_ops()
        x2 = x.view(-1, x.shape[-1])
        if x2.shape[0] <= SKINNY_MAX_M or x2.dtype == torch.float32 or not x2.is_cuda:
            y = ops.linear(x2, layer.weight, bias)
        else:
            y = F.linear(x2, layer.weight, bias)
        return y.view(*x.shape[:-1], y.shape[-1])
