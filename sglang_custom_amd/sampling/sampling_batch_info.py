"""Per-batch sampling tensors (python/sglang/srt/sampling/sampling_batch_info.py:69-160):
temperatures [B,1] f32, top_ps [B] f32, top_ks [B] i32, min_ps [B] f32;
is_all_greedy = every top_k <= 1 (:154)."""
import dataclasses
from typing import List

import torch


@dataclasses.dataclass
class SamplingBatchInfo:
    temperatures: torch.Tensor
    top_ps: torch.Tensor
    top_ks: torch.Tensor
    min_ps: torch.Tensor
    is_all_greedy: bool
    need_min_p_sampling: bool
    vocab_size: int
    device: str

    @classmethod
    def from_reqs(cls, reqs: List, vocab_size: int, device) -> "SamplingBatchInfo":
        sp = [r.sampling_params for r in reqs]
        return cls(
            temperatures=torch.tensor([p.temperature for p in sp], dtype=torch.float).view(-1, 1).to(device, non_blocking=True),
            top_ps=torch.tensor([p.top_p for p in sp], dtype=torch.float).to(device, non_blocking=True),
            top_ks=torch.tensor([p.top_k for p in sp], dtype=torch.int32).to(device, non_blocking=True),
            min_ps=torch.tensor([p.min_p for p in sp], dtype=torch.float).to(device, non_blocking=True),
            is_all_greedy=all(p.top_k <= 1 for p in sp),
            need_min_p_sampling=any(p.min_p > 0 for p in sp),
            vocab_size=vocab_size,
            device=device,
        )

    @classmethod
    def from_schedule_batch(cls, batch, vocab_size: int) -> "SamplingBatchInfo":
        return cls.from_reqs(batch.reqs, vocab_size, batch.device)

    def filter_batch(self, keep_indices: List[int], keep_indices_device: torch.Tensor) -> None:
        """Keep the rows of the surviving requests (sampling_batch_info.py filter_batch)."""
        self.temperatures = self.temperatures[keep_indices_device]
        self.top_ps = self.top_ps[keep_indices_device]
        self.top_ks = self.top_ks[keep_indices_device]
        self.min_ps = self.min_ps[keep_indices_device]

    def merge_batch(self, other: "SamplingBatchInfo") -> None:
        """Append another batch's rows (sampling_batch_info.py merge_batch)."""
        self.temperatures = torch.cat([self.temperatures, other.temperatures])
        self.top_ps = torch.cat([self.top_ps, other.top_ps])
        self.top_ks = torch.cat([self.top_ks, other.top_ks])
        self.min_ps = torch.cat([self.min_ps, other.min_ps])
        self.is_all_greedy = self.is_all_greedy and other.is_all_greedy
        self.need_min_p_sampling = self.need_min_p_sampling or other.need_min_p_sampling
