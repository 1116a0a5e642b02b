"""UnquantizedLinearMethod.apply(layer, x, bias) = F.linear
(python/sglang/srt/layers/quantization/unquant.py:110-122).

Decode-sized inputs (M <= 64 rows) and every fp32 input go to the hand-written kernels behind
mi355_linear (weight-streaming MFMA / exact-fp32 tiled).  Large-M bf16/f16 GEMMs (prefill) run on the
hand-written 256 x 256-tile matrix-core kernel behind mi355_linear_tiled, which can apply SiLU-and-mul in
its epilogue.  PREFILL_GEMM (env MI355_PREFILL_GEMM) picks: "tiled" (default) = that kernel for every
shape it serves (at least TILED_MIN_TILES output tiles; smaller problems and shapes it does not take
fall back to the library GEMM, torch.nn.functional.linear -> hipBLASLt); "auto" = the faster of the two
per projection as measured on MI355X (the hand-written kernel for gate_up with the fused activation,
1.02x the library pair; the library for the plain projections, where it is 4-6 % faster); "library" =
hipBLASLt everywhere."""
import os
from typing import Optional

import torch
import torch.nn.functional as F

from ...ops_backend import get_ops

SKINNY_MAX_M = 64
# Deferred split-K reduction / fused decode step (bit-identical to the unfused kernels; the switch
# exists so that tests and profiles can run both).
DECODE_FUSION = True
# "tiled" | "auto" | "library" for 16-bit GEMMs with more than SKINNY_MAX_M rows (see the module docstring)
PREFILL_GEMM = os.environ.get("MI355_PREFILL_GEMM", "tiled")
# below this many 256 x 256 output tiles the one-tile-per-workgroup kernel leaves most CUs idle
TILED_MIN_TILES = 192


def _use_tiled(m: int, n_out_tiles_cols: int, n: int, k: int, dtype, silu_mul: bool) -> bool:
    if PREFILL_GEMM == "library" or (PREFILL_GEMM == "auto" and not silu_mul):
        return False
    tiles = ((m + 255) // 256) * ((n_out_tiles_cols + 255) // 256)
    return tiles >= TILED_MIN_TILES and get_ops().linear_tiled_supported(m, n, k, dtype, silu_mul)


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
    w = layer.weight
    if (x2.is_cuda and isinstance(layer.quant_method, UnquantizedLinearMethod) and getattr(layer, "bias", None) is None
            and x2.stride(1) == 1 and x2.stride(0) % 8 == 0
            and _use_tiled(x2.shape[0], w.shape[0], w.shape[0], w.shape[1], x2.dtype, True)):
        y = get_ops().linear_tiled(x2, w, None, silu_mul=True)
        return y.view(*x.shape[:-1], y.shape[-1])
    gate_up, _ = layer(x)
    return act_fn(gate_up)


class SplitKPartials:
    """Output of a decode GEMM whose split-K partials ([splits, M, N] fp32) are still to be summed
    -- by the kernel that consumes it (the fused add + RMSNorm, or the fused decode attention).
    x_scale / w_scale: the fp8 GEMM's per-token and per-channel scales (None for a 16-bit GEMM)."""

    def __init__(self, part: torch.Tensor, x_scale=None, w_scale=None):
        self.part = part
        self.x_scale = x_scale
        self.w_scale = w_scale
        self.shape = part.shape[1:]


class QuantizedAct:
    """A decode activation that already went through per-token fp8 quantisation inside the kernel
    that produced it (row norm, SiLU-mul): xq [M, K] e4m3fn, x_scale [M, 1]; `dense` is the 16-bit
    tensor when the producer also wrote it (None otherwise)."""

    def __init__(self, xq: torch.Tensor, x_scale: torch.Tensor, dtype: torch.dtype,
                 dense: Optional[torch.Tensor] = None):
        self.xq = xq
        self.x_scale = x_scale
        self.dtype = dtype      # the activation dtype of the model (output dtype of the GEMM that follows)
        self.dense = dense
        self.shape = xq.shape


class PendingAllReduce:
    """This rank's partial sums of a row-parallel projection whose sum over the TP group has been
    left to the consumer: the RMSNorm behind it does all-reduce + residual add + norm in one launch
    (the reference defers the same all-reduce through LayerCommunicator, layers/communicator.py)."""

    def __init__(self, partial: torch.Tensor):
        self.partial = partial

    @property
    def shape(self):
        return self.partial.shape

    @property
    def dtype(self):
        return self.partial.dtype


def decode_linear_deferred(layer, x, needs_all_reduce: bool = True):
    """layer(x) on the decode path, leaving the split-K reduction to the kernel that consumes it; None
    when the layer, the shape or the ops backend does not support it.  A row-parallel layer
    (needs_all_reduce) qualifies only at TP=1: its output would have to cross ranks first.  x is a
    [M, K] tensor or, for an fp8 layer, a QuantizedAct."""
    from .w8a8_fp8 import W8A8Fp8LinearMethod

    ops = get_ops()
    wp = getattr(layer, "weight_packed", None)
    if (not DECODE_FUSION or wp is None or (needs_all_reduce and getattr(layer, "tp_size", 1) != 1)
            or len(x.shape) != 2 or not 1 <= x.shape[0] <= SKINNY_MAX_M):
        return None
    n = layer.weight.shape[0]
    if isinstance(layer.quant_method, W8A8Fp8LinearMethod):
        if isinstance(x, QuantizedAct):
            xq, xs = x.xq, x.x_scale
        else:
            xq, xs = ops.per_token_quant_fp8(x, pad_to=layer.k_padded)
        if xq.shape[1] != layer.k_padded:
            return None
        part = ops.fp8_packed_partials(xq, wp, n)
        return None if part is None else SplitKPartials(part, xs, layer.weight_scale.data)
    if (isinstance(x, QuantizedAct) or not isinstance(layer.quant_method, UnquantizedLinearMethod)
            or x.stride(0) % 8 != 0):
        return None
    part = ops.linear_packed_partials(x, wp, n)
    return None if part is None else SplitKPartials(part)


class UnquantizedLinearMethod:
    def process_weights_after_loading(self, layer) -> None:
        pack_for_decode(layer)

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        x2 = x.view(-1, x.shape[-1])
        if x2.shape[0] <= SKINNY_MAX_M or x2.dtype == torch.float32 or not x2.is_cuda:
            y = decode_linear(layer, x2, bias)
        elif (x2.stride(1) == 1 and x2.stride(0) % 8 == 0 and layer.weight.is_contiguous()
              and _use_tiled(x2.shape[0], layer.weight.shape[0], layer.weight.shape[0], layer.weight.shape[1],
                             x2.dtype, False)):
            y = get_ops().linear_tiled(x2, layer.weight, bias)
        else:
            y = F.linear(x2, layer.weight, bias)
        return y.view(*x.shape[:-1], y.shape[-1])
