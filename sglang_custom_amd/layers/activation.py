"""SiluAndMul (python/sglang/srt/layers/activation.py:55-65): x [.., 2d] -> [.., d]."""
import torch

from ..custom_op import CustomOp
from ..ops_backend import get_ops


class SiluAndMul(CustomOp):
    def forward_hip(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_contiguous():
            x = x.contiguous()
        return get_ops().silu_and_mul(x)
