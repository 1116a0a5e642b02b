"""RMSNorm with the reference's return conventions
(python/sglang/srt/layers/layernorm.py:59-164): forward(x) -> out;
forward(x, residual) -> (x, residual), both updated IN PLACE like forward_cuda/forward_hip
(:76-78, :121-123)."""
from typing import Optional, Tuple, Union

import torch
from torch import nn

from ..custom_op import CustomOp
from ..ops_backend import get_ops
from .quantization.unquant import SplitKPartials


class RMSNorm(CustomOp):
    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size), requires_grad=False)
        self.variance_epsilon = eps

    def forward_hip(
        self, x: torch.Tensor, residual: Optional[torch.Tensor] = None
    ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        ops = get_ops()
        if isinstance(x, SplitKPartials):
            # the producing GEMM left its split-K partials for this kernel to sum
            assert residual is not None
            return ops.fused_add_rmsnorm_partials(x.part, residual, self.weight.data, self.variance_epsilon), residual
        if not x.is_contiguous():
            x = x.contiguous()
        if residual is not None:
            ops.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        return ops.rmsnorm(x, self.weight.data, self.variance_epsilon)
