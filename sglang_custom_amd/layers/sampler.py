"""Sampler.forward (python/sglang/srt/layers/sampler.py:38-151): greedy -> argmax on fp32
logits; otherwise in-place temperature softmax then joint top-k / top-p / min-p sampling.

The non-greedy draw is sort-free (threshold search + inverse CDF in index order) with uniforms
from a torch.Generator the caller may seed; it samples from the same filtered, renormalised
distribution as top_k_top_p_min_p_sampling_from_probs_torch (:154-178), except that entries
exactly tied with the cut-off probability are all kept (the reference keeps an unspecified
subset of ties, decided by its unstable sort)."""
import os
from typing import Optional

import torch
from torch import nn

from ..ops_backend import get_ops

# The reference's switch, same name (layers/sampler.py:26): off by default -- every kernel on the path is
# deterministic and all ranks see the same gathered logits -- on as cheap insurance for a first multi-GPU run.
SYNC_TOKEN_IDS_ACROSS_TP = os.environ.get("SYNC_TOKEN_IDS_ACROSS_TP", "").lower() in ("1", "true", "yes")


class Sampler(nn.Module):
    def __init__(self, generator: Optional[torch.Generator] = None):
        super().__init__()
        self.generator = generator

    def forward(self, logits_output, sampling_info, return_logprob: bool = False,
                top_logprobs_nums=None, token_ids_logprobs=None) -> torch.Tensor:
        ops = get_ops()
        logits = logits_output.next_token_logits
        if not logits.is_contiguous():
            logits = logits.contiguous()
        if sampling_info.is_all_greedy:
            batch_next_token_ids = ops.argmax(logits)
            if return_logprob:
                logprobs = torch.nn.functional.log_softmax(logits, dim=-1)
        else:
            probs = ops.softmax_temperature_(logits, sampling_info.temperatures.view(-1))
            u = torch.rand(logits.shape[0], dtype=torch.float32, device=logits.device,
                           generator=self.generator)
            batch_next_token_ids = ops.top_k_top_p_min_p_sample(
                probs, sampling_info.top_ks, sampling_info.top_ps,
                sampling_info.min_ps if sampling_info.need_min_p_sampling else None, u)
            if return_logprob:
                logprobs = torch.log(probs).clamp(min=torch.finfo(probs.dtype).min)
        if return_logprob:
            idx = torch.arange(len(batch_next_token_ids), device=logits.device)
            logits_output.next_token_logprobs = logprobs[idx, batch_next_token_ids.long()]
        if SYNC_TOKEN_IDS_ACROSS_TP or getattr(sampling_info, "grammars", None):
            from ..distributed.parallel_state import sync_token_ids_across_tp

            batch_next_token_ids = sync_token_ids_across_tp(batch_next_token_ids)
        return batch_next_token_ids
