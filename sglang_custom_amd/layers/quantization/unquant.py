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
# Deferred split-K reduction / fused decode step (bit-identical to the unfused kernels; the switch
# exists so that tests and profiles can run both).
DECODE_FUSION = True


def pack_for_decode(layer) -> None:
    """Keep a second copy of a 16-bit weight in the decode GEMM's streaming order (MFMA-fragment
    blocks, ops.pack_weight).  The row-major copy stays: prefill GEMMs, the embedding gather and
    checkpoint loading use it.  2x weight bytes is the price of reading decode weights as whole
    contiguous lines; at 288 GB of HBM per GPU it is not the binding constraint."""
    w = layer.weight
    layer.weight_packed = None
    if w.is_cuda and w.dtype in (torch.bfloat16, torch.float16) and w.shape[1] % 32 == 0:
        layer.weight_packed = get_ops().pack_weight(w.data)


def decode_linear(layer, x2: torch.Tensor, bias: Optional[torch.Tensor], out_dtype=None) -> torch.Tensor:
    """M <= 64 rows (or fp32): the hand-written kernels, on the packed copy when there is one."""
    ops = get_ops()
    wp = getattr(layer, "weight_packed", None)
    if wp is not None and x2.shape[0] <= SKINNY_MAX_M and x2.stride(0) % 8 == 0:
        return ops.linear_packed(x2, wp, layer.weight.shape[0], bias, out_dtype=out_dtype)
    return ops.linear(x2, layer.weight, bias, out_dtype=out_dtype)


def pack_gate_up_for_decode(layer) -> None:
    """gate_up_proj: the decode copy interleaves gate and up rows (8 + 8 per MFMA tile) so that the
    GEMM epilogue can apply SiLU-and-mul in registers (ops.linear_packed_silu_mul).  It replaces
    the plain packed copy: the unfused GEMM + activation remains the path for large M."""
    w = layer.weight
    layer.weight_packed = None
    layer.weight_packed_gate_up = None
    if (w.is_cuda and w.dtype in (torch.bfloat16, torch.float16) and w.shape[1] % 256 == 0
            and w.shape[0] % 16 == 0 and w.shape[0] < 65536):
        layer.weight_packed_gate_up = get_ops().pack_weight_gate_up(w.data)
    else:
        pack_for_decode(layer)


def gate_up_silu_mul(layer, act_fn, x: torch.Tensor) -> torch.Tensor:
    """act_fn(gate_up_proj(x)) (models/llama.py:94-98), fused for decode-sized inputs."""
    x2 = x.view(-1, x.shape[-1])
    wp = getattr(layer, "weight_packed_gate_up", None)
    if wp is not None and x2.shape[0] <= SKINNY_MAX_M and x2.stride(0) % 8 == 0:
        y = get_ops().linear_packed_silu_mul(x2, wp, layer.weight.shape[0])
        return y.view(*x.shape[:-1], y.shape[-1])
    gate_up, _ = layer(x)
    return act_fn(gate_up)


class SplitKPartials:
    """Output of a row-parallel decode GEMM whose split-K partials ([splits, M, N] fp32) are still
    to be summed -- by the fused add + RMSNorm that consumes it (RMSNorm.forward accepts it as x)."""

    def __init__(self, part: torch.Tensor):
        self.part = part
        self.shape = part.shape[1:]


def decode_linear_deferred(layer, x: torch.Tensor, needs_all_reduce: bool = True):
    """layer(x) on the decode path, leaving the split-K reduction to the kernel that consumes it; None
    when the layer, the shape or the ops backend does not support it.  A row-parallel layer
    (needs_all_reduce) qualifies only at TP=1: its output would have to cross ranks first."""
    fn = getattr(get_ops(), "linear_packed_partials", None)
    wp = getattr(layer, "weight_packed", None)
    if (not DECODE_FUSION or fn is None or wp is None or (needs_all_reduce and getattr(layer, "tp_size", 1) != 1) or x.dim() != 2
            or not 1 <= x.shape[0] <= SKINNY_MAX_M or x.stride(0) % 8 != 0
            or not isinstance(layer.quant_method, UnquantizedLinearMethod)):
        return None
    part = fn(x, wp, layer.weight.shape[0])
    return None if part is None else SplitKPartials(part)


class UnquantizedLinearMethod:
    def process_weights_after_loading(self, layer) -> None:
        pack_for_decode(layer)

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        x2 = x.view(-1, x.shape[-1])
        if x2.shape[0] <= SKINNY_MAX_M or x2.dtype == torch.float32 or not x2.is_cuda:
            y = decode_linear(layer, x2, bias)
        else:
            y = F.linear(x2, layer.weight, bias)
        return y.view(*x.shape[:-1], y.shape[-1])
