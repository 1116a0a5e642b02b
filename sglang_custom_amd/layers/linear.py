"""Tensor-parallel linear layers with the reference's sharding rules
(python/sglang/srt/layers/linear.py: QKVParallelLinear :724-, MergedColumnParallelLinear :420-,
RowParallelLinear :1119-1303).  forward(x) returns (output, None) like the reference
(output_bias is always None here).  Weights are [out_per_rank, in_per_rank] row-major
(unquant.py:94-103)."""
from typing import List, Optional

import torch
from torch import nn

from ..distributed.parallel_state import (
    get_custom_all_reduce,
    get_tensor_model_parallel_rank,
    get_tensor_model_parallel_world_size,
    tensor_model_parallel_all_reduce,
)
from .quantization.unquant import PendingAllReduce, UnquantizedLinearMethod


class LinearBase(nn.Module):
    def __init__(self, in_per_rank: int, out_per_rank: int, dtype: torch.dtype, device,
                 quant_method=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_per_rank, in_per_rank, dtype=dtype, device=device),
                                   requires_grad=False)
        self.quant_method = quant_method or UnquantizedLinearMethod()
        self.tp_rank = get_tensor_model_parallel_rank()
        self.tp_size = get_tensor_model_parallel_world_size()


class ReplicatedLinear(LinearBase):
    """Unsharded projection (linear.py ReplicatedLinear): every rank holds the whole weight."""

    def __init__(self, input_size: int, output_size: int, dtype, device, quant_method=None):
        super().__init__(input_size, output_size, dtype, device, quant_method)

    def forward(self, x):
        return self.quant_method.apply(self, x), None


class QKVParallelLinear(LinearBase):
    """Fused q|k|v projection, column-parallel over heads.  Per rank the rows are
    [q heads | k heads | v heads] (linear.py:784-794); with fewer kv heads than ranks each kv head
    is replicated tp/Hkv times and rank r owns kv head r // replicas (:778-783)."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int, dtype, device, quant_method=None):
        tp = get_tensor_model_parallel_world_size()
        assert total_num_heads % tp == 0
        self.head_size = head_size
        self.num_heads = total_num_heads // tp
        if total_num_kv_heads >= tp:
            assert total_num_kv_heads % tp == 0
            self.num_kv_heads, self.num_kv_head_replicas = total_num_kv_heads // tp, 1
        else:
            assert tp % total_num_kv_heads == 0
            self.num_kv_heads, self.num_kv_head_replicas = 1, tp // total_num_kv_heads
        out = (self.num_heads + 2 * self.num_kv_heads) * head_size
        super().__init__(hidden_size, out, dtype, device, quant_method)

    def shard_from_checkpoint(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """Full q_proj/k_proj/v_proj weights -> this rank's fused [q|k|v] rows."""
        hs, r = self.head_size, self.tp_rank
        qs = q[r * self.num_heads * hs:(r + 1) * self.num_heads * hs]
        kv_rank = r // self.num_kv_head_replicas
        ks = k[kv_rank * self.num_kv_heads * hs:(kv_rank + 1) * self.num_kv_heads * hs]
        vs = v[kv_rank * self.num_kv_heads * hs:(kv_rank + 1) * self.num_kv_heads * hs]
        return torch.cat([qs, ks, vs], dim=0)

    def forward(self, x):
        return self.quant_method.apply(self, x), None


class MergedColumnParallelLinear(LinearBase):
    """gate|up projection: each output block is column-sharded, rank rows are [gate | up]."""

    def __init__(self, input_size: int, output_sizes: List[int], dtype, device, quant_method=None):
        tp = get_tensor_model_parallel_world_size()
        assert all(o % tp == 0 for o in output_sizes)
        self.output_sizes = output_sizes
        super().__init__(input_size, sum(output_sizes) // tp, dtype, device, quant_method)

    def shard_from_checkpoint(self, *blocks: torch.Tensor) -> torch.Tensor:
        parts = []
        for w in blocks:
            per = w.shape[0] // self.tp_size
            parts.append(w[self.tp_rank * per:(self.tp_rank + 1) * per])
        return torch.cat(parts, dim=0)

    def forward(self, x):
        return self.quant_method.apply(self, x), None


class RowParallelLinear(LinearBase):
    """Input-sharded projection; partial outputs are summed over the TP group
    (linear.py:1296-1297)."""

    def __init__(self, input_size: int, output_size: int, dtype, device, quant_method=None,
                 reduce_results: bool = True):
        tp = get_tensor_model_parallel_world_size()
        assert input_size % tp == 0
        self.reduce_results = reduce_results
        super().__init__(input_size // tp, output_size, dtype, device, quant_method)

    def shard_from_checkpoint(self, w: torch.Tensor) -> torch.Tensor:
        per = w.shape[1] // self.tp_size
        return w[:, self.tp_rank * per:(self.tp_rank + 1) * per].contiguous()

    def forward(self, x, defer_all_reduce: bool = False):
        """defer_all_reduce: the caller hands the result to an RMSNorm with a residual; when the
        one-launch collectives are on and the output qualifies, the partial sums come back as a
        PendingAllReduce for that norm to sum (all-reduce + add + norm in one launch)."""
        y = self.quant_method.apply(self, x)
        if self.reduce_results and self.tp_size > 1:
            ca = get_custom_all_reduce() if defer_all_reduce else None
            if ca is not None and y.is_cuda and ca.should_fuse_norm(y, y.shape[-1]):
                return PendingAllReduce(y), None
            y = tensor_model_parallel_all_reduce(y)
        return y, None
