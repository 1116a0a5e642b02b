"""fp8 W8A8 linear: per-channel weight scales, dynamic per-token activation scales, OCP
e4m3fn (gfx950; not the gfx942 fnuz flavour -- reference logic fp8_kernel.py:71-85).

Mirrors W8A8Fp8LinearMethod (python/sglang/srt/layers/quantization/w8a8_fp8.py:98-189) and the
non-CUDA branch of apply_fp8_linear (fp8_utils.py:676-749): quantise activations per token
(group = K), fp8 x fp8 -> fp32 GEMM, then * x_scale * w_scale (+ bias) -> input dtype.
The weight stays [N, K] (one contiguous row per output channel), which is the layout the
weight-streaming kernel wants; the reference stores the transpose for torch._scaled_mm
(w8a8_fp8.py:115,132)."""
from typing import Optional

import torch

from ...ops_backend import get_ops

FP8_DTYPE = torch.float8_e4m3fn
FP8_MAX = 448.0


def quantize_weight_per_channel(w: torch.Tensor):
    """[N, K] high-precision weight -> (fp8 [N, K], scale [N, 1] fp32); what the reference's
    process_weights_after_loading does for non-fp8 checkpoints (w8a8_fp8.py:118-134 via
    per_token_group_quant_fp8 with group = K)."""
    wf = w.float()
    scale = wf.abs().amax(dim=1, keepdim=True).clamp(min=1e-10) / FP8_MAX
    q = (wf * (1.0 / scale)).clamp(-FP8_MAX, FP8_MAX).to(FP8_DTYPE)
    return q, scale


class W8A8Fp8LinearMethod:
    def process_weights_after_loading(self, layer) -> None:
        if layer.weight.dtype != FP8_DTYPE:
            q, s = quantize_weight_per_channel(layer.weight.data)
            layer.weight = torch.nn.Parameter(q, requires_grad=False)
            layer.weight_scale = torch.nn.Parameter(s.contiguous(), requires_grad=False)

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        ops = get_ops()
        x2 = x.view(-1, x.shape[-1])
        xq, xs = ops.per_token_quant_fp8(x2)
        y = ops.fp8_scaled_mm(xq, layer.weight, xs, layer.weight_scale, x.dtype, bias)
        return y.view(*x.shape[:-1], y.shape[-1])
