"""RMSNorm with the reference's return conventions
(python/sglang/srt/layers/layernorm.py:59-164): forward(x) -> out;
forward(x, residual) -> (x, residual), both updated IN PLACE like forward_cuda/forward_hip
(:76-78, :121-123)."""
from typing import Optional, Tuple, Union

import torch
from torch import nn

from ..custom_op import CustomOp
from ..ops_backend import get_ops
from ..distributed.parallel_state import get_custom_all_reduce, tensor_model_parallel_all_reduce
from .quantization.unquant import PendingAllReduce, QuantizedAct, SplitKPartials


class RMSNorm(CustomOp):
    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size), requires_grad=False)
        self.variance_epsilon = eps

    def forward_hip(
        self, x, residual: Optional[torch.Tensor] = None, quant_fp8: bool = False
    ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        """x: a tensor, or the SplitKPartials of the GEMM that produced it (summed here).
        quant_fp8: also quantise the normalised rows per token for an fp8 GEMM -- the result is then
        a QuantizedAct (decode-sized inputs only; callers check the ops backend supports it)."""
        ops = get_ops()
        if isinstance(x, PendingAllReduce):
            # partial sums of a row-parallel projection (layernorm.py:166-190 forward_with_allreduce_fusion)
            ca = get_custom_all_reduce()
            if residual is not None and not quant_fp8 and ca is not None:
                ca.fused_add_rmsnorm(x.partial, residual, self.weight.data, self.variance_epsilon)
                return x.partial, residual
            x = tensor_model_parallel_all_reduce(x.partial)
        if isinstance(x, SplitKPartials) or quant_fp8:
            part = x if isinstance(x, SplitKPartials) else None
            assert part is None or residual is not None
            if part is not None and not quant_fp8 and part.x_scale is None:
                return ops.fused_add_rmsnorm_partials(part.part, residual, self.weight.data, self.variance_epsilon), residual
            dense = None if part is not None else (x if x.is_contiguous() else x.contiguous())
            out, xq, xs = ops.rmsnorm_ex(self.weight.data, self.variance_epsilon, x=dense, residual=residual,
                                         part=part.part if part is not None else None,
                                         part_x_scale=part.x_scale if part is not None else None,
                                         part_w_scale=part.w_scale if part is not None else None,
                                         quant_fp8=quant_fp8)
            y = QuantizedAct(xq, xs, out.dtype, out) if quant_fp8 else out
            return (y, residual) if residual is not None else y
        if not x.is_contiguous():
            x = x.contiguous()
        if residual is not None:
            ops.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        return ops.rmsnorm(x, self.weight.data, self.variance_epsilon)
