"""fp8 W8A8 linear: per-channel weight scales, dynamic per-token activation scales, OCP
e4m3fn (gfx950; not the gfx942 fnuz flavour -- reference logic fp8_kernel.py:71-85).

Mirrors W8A8Fp8LinearMethod (python/sglang/srt/layers/quantization/w8a8_fp8.py:98-189) and the
non-CUDA branch of apply_fp8_linear (fp8_utils.py:676-749): quantise activations per token
(group = K), fp8 x fp8 -> fp32 GEMM, then * x_scale * w_scale (+ bias) -> input dtype.
The weight stays [N, K] (one contiguous row per output channel), which is the layout the
weight-streaming kernel wants; the reference stores the transpose for torch._scaled_mm
(w8a8_fp8.py:115,132)."""
import os
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


SKINNY_MAX_M = 64
# Large-M fp8 GEMMs: "auto" (default) and "tiled" = the hand-written 256 x 256-tile kernel
# (mi355_fp8_scaled_mm_tiled) for every shape it serves -- on MI355X it is 1.00-1.10x torch._scaled_mm
# (hipBLASLt) on the plain projections and 1.23x on gate_up, whose scales and SiLU-and-mul run in its
# epilogue; "library" = torch._scaled_mm everywhere.
PREFILL_GEMM = os.environ.get("MI355_PREFILL_FP8_GEMM", "auto")
TILED_MIN_TILES = 192   # below this many 256 x 256 output tiles most CUs would idle
_SCALED_MM_OK = None   # None = untried; torch._scaled_mm (hipBLASLt) with row-wise scales


def _library_scaled_mm(xq, wq, xs, ws, out_dtype, bias):
    """Large-M fp8 GEMM as a plain library call: torch._scaled_mm with per-row / per-column
    scales, the call the reference itself makes (fp8_utils.py:720-749).  Returns None when this
    torch build does not support it on the device."""
    global _SCALED_MM_OK
    if _SCALED_MM_OK is False:
        return None
    try:
        y = torch._scaled_mm(xq, wq.t(), scale_a=xs, scale_b=ws.t(), out_dtype=out_dtype, bias=bias)
        _SCALED_MM_OK = True
        return y
    except (RuntimeError, ValueError, NotImplementedError):
        if _SCALED_MM_OK:   # it worked before: a real error, not a capability probe
            raise
        _SCALED_MM_OK = False
        return None


def _tiled_ok(m: int, n: int, k: int, out_dtype, silu_mul: bool) -> bool:
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    return tiles >= TILED_MIN_TILES and get_ops().fp8_scaled_mm_tiled_supported(m, n, k, out_dtype, silu_mul)


def gate_up_silu_mul_prefill(layer, x: torch.Tensor) -> Optional[torch.Tensor]:
    """act_fn(gate_up_proj(x)) (models/llama.py:94-98) for a prefill-sized fp8 W8A8 gate_up
    projection: per-token quantisation, then ONE launch that multiplies, applies the scales and
    SiLU-and-mul.  None when the shape or the switch says to take the unfused path."""
    if not isinstance(x, torch.Tensor):   # a QuantizedAct: the decode path, served by the packed kernels
        return None
    x2 = x.view(-1, x.shape[-1])
    w = layer.weight
    if (PREFILL_GEMM == "library" or not x2.is_cuda or x2.shape[0] <= SKINNY_MAX_M
            or getattr(layer, "bias", None) is not None or w.dtype != FP8_DTYPE
            or not _tiled_ok(x2.shape[0], w.shape[0], w.shape[1], x2.dtype, True)):
        return None
    ops = get_ops()
    xq, xs = ops.per_token_quant_fp8(x2)
    y = ops.fp8_scaled_mm_tiled(xq, w, xs, layer.weight_scale.view(-1), x2.dtype, None, silu_mul=True)
    return y.view(*x.shape[:-1], y.shape[-1])


class W8A8Fp8LinearMethod:
    def process_weights_after_loading(self, layer) -> None:
        if layer.weight.dtype != FP8_DTYPE:
            q, s = quantize_weight_per_channel(layer.weight.data)
            layer.weight = torch.nn.Parameter(q, requires_grad=False)
            layer.weight_scale = torch.nn.Parameter(s.contiguous(), requires_grad=False)
        # decode copy in the weight-streaming kernel's fragment order (the row-major one stays for
        # the large-M library GEMM)
        # A reduction length the kernel does not stream (K % 512 != 0: DeepSeek-V3's down_proj shard, 18432 / 8 =
        # 2304) is rounded up with zero weights; the activation quantisers write the matching zero tail
        # (k_padded is the row length the packed GEMM expects of xq).
        w = layer.weight
        layer.weight_packed = None
        n, k = w.shape
        layer.k_padded = k
        if w.is_cuda and k % 16 == 0 and n < 65536:
            kp = -(-k // 512) * 512
            if kp != k:
                wp = torch.zeros((n, kp), dtype=torch.uint8, device=w.device)
                wp[:, :k] = w.data.view(torch.uint8)
                w = wp.view(FP8_DTYPE)
            layer.weight_packed = get_ops().pack_weight_fp8(w.data)
            layer.k_padded = kp

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        from .unquant import QuantizedAct

        ops = get_ops()
        if isinstance(x, QuantizedAct):   # quantised by the kernel that produced it (decode path)
            xq, xs = x.xq, x.x_scale
            out_dtype = x.dtype
            wp = getattr(layer, "weight_packed", None)
            if wp is not None and xq.shape[0] <= SKINNY_MAX_M and xq.shape[1] == layer.k_padded:
                return ops.fp8_scaled_mm_packed(xq, wp, layer.weight.shape[0], xs, layer.weight_scale, out_dtype, bias)
            assert xq.shape[1] == layer.weight.shape[1], "a padded activation is for the packed decode GEMM only"
            return ops.fp8_scaled_mm(xq, layer.weight, xs, layer.weight_scale, out_dtype, bias)
        x2 = x.view(-1, x.shape[-1])
        wp = getattr(layer, "weight_packed", None)
        if wp is not None and x2.shape[0] <= SKINNY_MAX_M and x2.is_cuda:   # decode: the packed kernel (K padded if need be)
            xq, xs = ops.per_token_quant_fp8(x2, pad_to=layer.k_padded)
            y = ops.fp8_scaled_mm_packed(xq, wp, layer.weight.shape[0], xs, layer.weight_scale, x.dtype, bias)
            return y.view(*x.shape[:-1], y.shape[-1])
        xq, xs = ops.per_token_quant_fp8(x2)
        y = None
        if (PREFILL_GEMM != "library" and x2.shape[0] > SKINNY_MAX_M and x2.is_cuda
                and _tiled_ok(x2.shape[0], layer.weight.shape[0], x2.shape[1], x2.dtype, False)):
            y = ops.fp8_scaled_mm_tiled(xq, layer.weight, xs, layer.weight_scale.view(-1), x.dtype, bias)
        if (y is None and x2.shape[0] > SKINNY_MAX_M and x2.is_cuda and x2.shape[1] % 16 == 0
                and layer.weight.shape[0] % 16 == 0):
            y = _library_scaled_mm(xq, layer.weight, xs, layer.weight_scale, x.dtype, bias)
        if y is None:
            y = ops.fp8_scaled_mm(xq, layer.weight, xs, layer.weight_scale, x.dtype, bias)
        return y.view(*x.shape[:-1], y.shape[-1])
