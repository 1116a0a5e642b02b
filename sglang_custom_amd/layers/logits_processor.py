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
    # prefill with return_logprob: log-probability of every kept prompt token given its prefix, and the
    # top-k rows per request (logits_processor.py:64-77)
    input_token_logprobs: Optional[torch.Tensor] = None
    input_top_logprobs_val: Optional[list] = None
    input_top_logprobs_idx: Optional[list] = None


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
        if (forward_batch.forward_mode.is_extend() and getattr(forward_batch, "return_logprob", False)
                and forward_batch.extend_logprob_start_lens_cpu is not None
                and any(n - s > 0 for n, s in zip(forward_batch.extend_seq_lens_cpu,
                                                  forward_batch.extend_logprob_start_lens_cpu))):
            return self._forward_with_input_logprobs(hidden_states, lm_head, forward_batch)
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

    def _forward_with_input_logprobs(self, hidden_states, lm_head, fb) -> LogitsProcessorOutput:
        """The extend_return_logprob branch (logits_processor.py:274-420): logits of every prompt row from
        extend_logprob_start_len on (at least the last row, which samples), log-softmax over the rows that
        report a prompt token, the given next token's value per row, and per-request top-k."""
        sample_pt, in_pt = -1, 0
        sample_indices, input_indices, pruned_rows, pruned_lens = [], [], [], []
        pt = 0
        for start, n in zip(fb.extend_logprob_start_lens_cpu, fb.extend_seq_lens_cpu):
            s0 = start - 1 if n == start else start      # a chunk that reports nothing still samples one token
            assert n > s0
            pruned_rows.append(torch.arange(pt + s0, pt + n, dtype=torch.int64))
            pt += n
            sample_pt += n - s0
            sample_indices.append(sample_pt)
            input_indices.extend(in_pt + i for i in range(n - start))
            in_pt += n - s0
            pruned_lens.append(n - start)
        dev = hidden_states.device
        rows = torch.cat(pruned_rows).to(dev)
        pruned = get_ops().embedding(rows, hidden_states.contiguous())     # row gather
        logits = self._get_logits(pruned, lm_head)
        if self.final_logit_softcapping:
            if not logits.is_contiguous():
                logits = logits.contiguous()
            get_ops().softcap_inplace(logits, self.final_logit_softcapping)
        sampled = logits[torch.tensor(sample_indices, dtype=torch.int64, device=dev)]
        input_logits = logits[torch.tensor(input_indices, dtype=torch.int64, device=dev)]
        logprobs = torch.nn.functional.log_softmax(input_logits, dim=-1)
        out = LogitsProcessorOutput(next_token_logits=sampled)
        ids = fb.extend_input_logprob_token_ids_gpu
        out.input_token_logprobs = logprobs[torch.arange(logprobs.shape[0], device=dev), ids]
        nums = fb.top_logprobs_nums or []
        if any(k > 0 for k in nums):
            top = logprobs.topk(max(nums), dim=1)
            vals, idxs = top.values.tolist(), top.indices.tolist()
            out.input_top_logprobs_val, out.input_top_logprobs_idx, p0 = [], [], 0
            for k, plen in zip(nums, pruned_lens):
                if plen <= 0:
                    out.input_top_logprobs_val.append([])
                    out.input_top_logprobs_idx.append([])
                    continue
                out.input_top_logprobs_val.append([vals[p0 + j][:k] for j in range(plen)])
                out.input_top_logprobs_idx.append([idxs[p0 + j][:k] for j in range(plen)])
                p0 += plen
        return out

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
