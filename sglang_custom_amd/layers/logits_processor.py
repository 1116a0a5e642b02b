"""LogitsProcessor: last-token gather -> lm_head GEMM -> (TP all-gather) -> fp32 logits.
Mirrors python/sglang/srt/layers/logits_processor.py:230-270 (which rows), :453-455 (matmul in
the weight dtype), :464-477 (all-gather), :490 (slice to vocab, .float()), :217-224 and :492-493
(final_logit_softcapping: negative = off; applied to the fp32 logits in place)."""
import dataclasses
from typing import Optional

import torch
from torch import nn

from ..distributed.parallel_state import (
    get_tensor_model_parallel_world_size,
    tensor_model_parallel_all_gather,
)
from ..ops_backend import get_ops
from .quantization.unquant import decode_linear


@dataclasses.dataclass
class LogitsProcessorOutput:
    # fields used on this path of the reference's LogitsProcessorOutput (logits_processor.py:50-77)
    next_token_logits: torch.Tensor
    hidden_states: Optional[torch.Tensor] = None
    next_token_logprobs: Optional[torch.Tensor] = None
    next_token_top_logprobs_val: Optional[list] = None
    next_token_top_logprobs_idx: Optional[list] = None


class LogitsProcessor(nn.Module):
    def __init__(self, config, logit_scale: Optional[float] = None):
        super().__init__()
        self.config = config
        self.logit_scale = logit_scale
        self.do_tensor_parallel_all_gather = get_tensor_model_parallel_world_size() > 1
        cap = getattr(config, "final_logit_softcapping", None)
        self.final_logit_softcapping = None if (cap is not None and cap < 0) else cap

    def forward(self, input_ids, hidden_states, lm_head, forward_batch) -> LogitsProcessorOutput:
        ops = get_ops()
        if forward_batch.forward_mode.is_decode_or_idle():
            pruned = hidden_states
        else:
            # last token of every sequence: cumsum(extend_seq_lens) - 1 (:256)
            last_index = torch.cumsum(forward_batch.extend_seq_lens, dim=0, dtype=torch.int64) - 1
            pruned = ops.embedding(last_index, hidden_states.contiguous())  # row gather
        logits = self._get_logits(pruned, lm_head)
        if self.final_logit_softcapping:
            if not logits.is_contiguous():   # a vocab-sliced view of padded logits
                logits = logits.contiguous()
            ops.softcap_inplace(logits, self.final_logit_softcapping)
        return LogitsProcessorOutput(next_token_logits=logits)

    def _get_logits(self, hidden_states: torch.Tensor, lm_head) -> torch.Tensor:
        w = lm_head.weight
        single = not self.do_tensor_parallel_all_gather and self.logit_scale is None
        x = hidden_states.to(w.dtype)
        if single and (x.shape[0] <= 64 or x.dtype == torch.float32 or not x.is_cuda):
            # T-rounded logits widened to fp32 in the GEMM epilogue (= matmul(...).float())
            logits = decode_linear(lm_head, x, None, out_dtype=torch.float32)
            return logits[:, : self.config.vocab_size]
        if x.shape[0] <= 64 or x.dtype == torch.float32 or not x.is_cuda:
            logits = decode_linear(lm_head, x, None)
        else:
            logits = torch.matmul(x, w.T)
        if self.logit_scale is not None:
            logits.mul_(self.logit_scale)
        if self.do_tensor_parallel_all_gather:
            logits = tensor_model_parallel_all_gather(logits)
        return logits[:, : self.config.vocab_size].float()
