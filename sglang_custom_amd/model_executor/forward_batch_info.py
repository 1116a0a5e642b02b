"""ForwardMode / ForwardBatch: the per-forward input record
(python/sglang/srt/model_executor/forward_batch_info.py:61-130, 155-428), field names and
dtypes as the reference passes them:

  input_ids i64 [T]; req_pool_indices i64 [B]; seq_lens i64 [B]; out_cache_loc i64 [T];
  positions i64 [T]; extend_seq_lens / extend_prefix_lens / extend_start_loc i32 [B] (+ _cpu lists)
"""
import dataclasses
from enum import IntEnum, auto
from typing import List, Optional

import torch

from ..ops_backend import get_ops


class ForwardMode(IntEnum):
    EXTEND = auto()         # 1
    DECODE = auto()         # 2
    MIXED = auto()          # 3: chunked prefill + decode requests as extends of length 1
    IDLE = auto()           # 4
    SPLIT_PREFILL = auto()  # 5
    TARGET_VERIFY = auto()  # 6
    DRAFT_EXTEND = auto()   # 7
    DUMMY_FIRST = auto()    # 8

    def is_prefill(self):
        return self.is_extend()

    def is_extend(self):
        return self in (ForwardMode.EXTEND, ForwardMode.MIXED, ForwardMode.DRAFT_EXTEND,
                        ForwardMode.TARGET_VERIFY)

    def is_decode(self):
        return self == ForwardMode.DECODE

    def is_mixed(self):
        return self == ForwardMode.MIXED

    def is_idle(self):
        return self == ForwardMode.IDLE

    def is_decode_or_idle(self):
        return self in (ForwardMode.DECODE, ForwardMode.IDLE)

    def is_cuda_graph(self):
        return self in (ForwardMode.DECODE, ForwardMode.TARGET_VERIFY, ForwardMode.IDLE)


def clamp_position(seq_lens: torch.Tensor) -> torch.Tensor:
    return torch.clamp(seq_lens - 1, min=0).to(torch.int64)


@dataclasses.dataclass
class ForwardBatch:
    forward_mode: ForwardMode
    batch_size: int
    input_ids: torch.Tensor
    req_pool_indices: torch.Tensor
    seq_lens: torch.Tensor
    out_cache_loc: torch.Tensor
    seq_lens_sum: int
    seq_lens_cpu: Optional[torch.Tensor] = None
    return_logprob: bool = False
    positions: torch.Tensor = None
    extend_num_tokens: Optional[int] = None
    extend_seq_lens: Optional[torch.Tensor] = None
    extend_prefix_lens: Optional[torch.Tensor] = None
    extend_start_loc: Optional[torch.Tensor] = None
    extend_prefix_lens_cpu: Optional[List[int]] = None
    extend_seq_lens_cpu: Optional[List[int]] = None
    sampling_info: object = None
    # input-token log-probabilities (forward_batch_info.py:183-190)
    top_logprobs_nums: Optional[List[int]] = None
    extend_logprob_start_lens_cpu: Optional[List[int]] = None
    extend_input_logprob_token_ids_gpu: Optional[torch.Tensor] = None
    req_to_token_pool: object = None
    token_to_kv_pool: object = None
    attn_backend: object = None

    @classmethod
    def init_new(cls, batch, model_runner) -> "ForwardBatch":
        """batch: ModelWorkerBatch (forward_batch_info.py:292-428)."""
        ret = cls(
            forward_mode=batch.forward_mode,
            batch_size=len(batch.seq_lens),
            input_ids=batch.input_ids,
            req_pool_indices=batch.req_pool_indices,
            seq_lens=batch.seq_lens,
            out_cache_loc=batch.out_cache_loc,
            seq_lens_sum=batch.seq_lens_sum,
            seq_lens_cpu=batch.seq_lens_cpu,
            return_logprob=batch.return_logprob,
            sampling_info=batch.sampling_info,
            req_to_token_pool=model_runner.req_to_token_pool,
            token_to_kv_pool=model_runner.token_to_kv_pool,
            attn_backend=model_runner.attn_backend,
        )
        device = model_runner.device
        if ret.forward_mode.is_idle():
            ret.positions = torch.empty((0,), device=device, dtype=torch.int64)
            return ret
        if ret.forward_mode.is_decode():
            ret.positions = batch.positions if batch.positions is not None else clamp_position(batch.seq_lens)
        else:
            ret.extend_seq_lens = torch.tensor(batch.extend_seq_lens, dtype=torch.int32).to(device, non_blocking=True)
            ret.extend_prefix_lens = torch.tensor(batch.extend_prefix_lens, dtype=torch.int32).to(device, non_blocking=True)
            ret.extend_num_tokens = batch.extend_num_tokens
            ret.positions, ret.extend_start_loc = get_ops().compute_position(
                ret.extend_prefix_lens, ret.extend_seq_lens, ret.extend_num_tokens)
            ret.extend_prefix_lens_cpu = batch.extend_prefix_lens
            ret.extend_seq_lens_cpu = batch.extend_seq_lens
            ret.top_logprobs_nums = getattr(batch, "top_logprobs_nums", None)
            ret.extend_logprob_start_lens_cpu = getattr(batch, "extend_logprob_start_lens", None)
            ids = getattr(batch, "extend_input_logprob_token_ids", None)
            if ids is not None:
                ret.extend_input_logprob_token_ids_gpu = ids.to(device, non_blocking=True)
        return ret
