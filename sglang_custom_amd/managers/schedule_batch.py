"""Req / ScheduleBatch / ModelWorkerBatch: CPU -> device batch materialisation and KV-slot
bookkeeping for one static batch, with the semantics of the reference's
python/sglang/srt/managers/schedule_batch.py (prepare_for_extend :1106-1303,
prepare_for_decode :1503-1573, get_model_worker_batch :1671-1756, ModelWorkerBatch :1829-1897).

What differs on purpose: the req->token table writes go through the HIP kernels
(mi355_write_req_to_token / mi355_prepare_decode) instead of Triton / index_put, the decode
update is ONE launch that also produces `positions`, and the sequence lengths are mirrored on
the host so that the attention planner never has to read a device tensor back.
"""
import dataclasses
from typing import List, Optional

import torch

from ..model_executor.forward_batch_info import ForwardMode
from ..ops_backend import get_ops
from ..sampling.sampling_batch_info import SamplingBatchInfo
from ..sampling.sampling_params import SamplingParams


class Req:
    """The request fields the hot path reads (schedule_batch.py Req)."""

    def __init__(self, rid, origin_input_ids: List[int], sampling_params: Optional[SamplingParams] = None,
                 origin_input_text: str = "", return_logprob: bool = False, top_logprobs_num: int = 0,
                 logprob_start_len: Optional[int] = None):
        self.rid = rid
        # log-probabilities (schedule_batch.py Req :473-510): of the prompt tokens from position
        # logprob_start_len on (default: none of them, only the sampled tokens), with the top-k of each row
        self.return_logprob = return_logprob
        self.top_logprobs_num = top_logprobs_num
        self.logprob_start_len = len(origin_input_ids) - 1 if logprob_start_len is None else logprob_start_len
        self.extend_logprob_start_len = 0
        self.origin_input_text = origin_input_text
        self.origin_input_ids = list(origin_input_ids)
        self.output_ids: List[int] = []
        self.sampling_params = sampling_params or SamplingParams(temperature=0)
        self.req_pool_idx: Optional[int] = None
        # prefix reuse: slots of an already-cached prefix (empty without a prefix cache)
        self.prefix_indices = []
        self.fill_ids = self.origin_input_ids
        self.extend_input_len = len(self.fill_ids)
        self.cached_tokens = 0
        self.already_computed = 0
        self.finished = False
        self.last_node = None   # radix-tree node its cached prefix ends in (locked while the request runs)
        self.is_retracted = False
        self.is_chunked = 0   # > 0 while a prompt is being prefilled in chunks (schedule_batch.py Req.is_chunked)

    def init_next_round_input(self, prefix_indices=None):
        """(Req.init_next_round_input) tokens to run next = prompt + outputs so far; the part that
        already has KV slots comes in as prefix_indices."""
        self.fill_ids = self.origin_input_ids + self.output_ids
        self.prefix_indices = [] if prefix_indices is None else prefix_indices
        self.extend_input_len = len(self.fill_ids) - len(self.prefix_indices)

    def reset_for_retract(self):
        """(Req.reset_for_retract) back to a waiting request that keeps its outputs so far."""
        self.prefix_indices = []
        self.last_node = None
        self.req_pool_idx = None
        self.extend_input_len = 0
        self.already_computed = 0
        self.is_chunked = 0
        self.is_retracted = True

    @property
    def seqlen(self) -> int:
        return len(self.origin_input_ids) + len(self.output_ids)


@dataclasses.dataclass
class ModelWorkerBatch:
    forward_mode: ForwardMode
    input_ids: torch.Tensor
    req_pool_indices: torch.Tensor
    seq_lens: torch.Tensor
    out_cache_loc: torch.Tensor
    seq_lens_sum: int
    seq_lens_cpu: Optional[torch.Tensor]
    return_logprob: bool
    extend_num_tokens: Optional[int]
    extend_seq_lens: Optional[List[int]]
    extend_prefix_lens: Optional[List[int]]
    sampling_info: SamplingBatchInfo
    positions: Optional[torch.Tensor] = None  # decode: produced by the prepare kernel
    # input-token log-probabilities of an extend batch (schedule_batch.py ModelWorkerBatch :1862-1866)
    top_logprobs_nums: Optional[List[int]] = None
    extend_logprob_start_lens: Optional[List[int]] = None
    extend_input_logprob_token_ids: Optional[torch.Tensor] = None


class ScheduleBatch:
    def __init__(self, reqs, req_to_token_pool, token_to_kv_pool_allocator, model_config, device):
        self.reqs: List[Req] = reqs
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool_allocator = token_to_kv_pool_allocator
        self.model_config = model_config
        self.device = device
        self.forward_mode: Optional[ForwardMode] = None
        self.return_logprob = any(getattr(r, "return_logprob", False) for r in reqs)   # schedule_batch.py:919
        self.top_logprobs_nums: Optional[List[int]] = None
        self.extend_logprob_start_lens: Optional[List[int]] = None
        self.extend_input_logprob_token_ids: Optional[torch.Tensor] = None
        self.input_ids = None
        self.req_pool_indices = None
        self.seq_lens = None
        self.seq_lens_cpu_list: List[int] = []
        self.out_cache_loc = None
        self.output_ids = None
        self.positions = None
        self.seq_lens_sum = 0
        self.sampling_info = None
        self.prefix_lens = None
        self.extend_lens = None
        self.extend_num_tokens = None

    @classmethod
    def init_new(cls, reqs, req_to_token_pool, token_to_kv_pool_allocator, tree_cache=None,
                 model_config=None, enable_overlap: bool = False, spec_algorithm=None,
                 enable_custom_logit_processor: bool = False):
        """Same keyword list as the reference's ScheduleBatch.init_new (schedule_batch.py:891)."""
        return cls(reqs, req_to_token_pool, token_to_kv_pool_allocator, model_config,
                   req_to_token_pool.device)

    def batch_size(self) -> int:
        return len(self.reqs)

    # -- allocation helpers (schedule_batch.py alloc_req_slots / alloc_token_slots) --
    def alloc_req_slots(self, num_reqs: int) -> List[int]:
        idx = self.req_to_token_pool.alloc(num_reqs)
        if idx is None:
            raise RuntimeError("alloc_req_slots runs out of memory: raise max_running_requests")
        return idx

    def alloc_token_slots(self, num_tokens: int) -> torch.Tensor:
        loc = self.token_to_kv_pool_allocator.alloc(num_tokens)
        if loc is None:
            raise RuntimeError(
                f"Out of KV slots: need {num_tokens}, "
                f"available {self.token_to_kv_pool_allocator.available_size()}")
        return loc

    def prepare_for_extend(self):
        self.forward_mode = ForwardMode.EXTEND
        bs = len(self.reqs)
        req_pool_indices = self.alloc_req_slots(bs)
        reqs = self.reqs
        input_ids = [r.fill_ids[len(r.prefix_indices):] for r in reqs]
        extend_num_tokens = sum(len(ids) for ids in input_ids)
        seq_lens = [len(r.fill_ids) for r in reqs]
        prefix_lens = [len(r.prefix_indices) for r in reqs]
        extend_lens = [r.extend_input_len for r in reqs]
        dev = self.device

        flat = [t for ids in input_ids for t in ids]
        self.req_pool_indices = torch.tensor(req_pool_indices, dtype=torch.int64).to(dev, non_blocking=True)
        self.input_ids = torch.tensor(flat, dtype=torch.int64).to(dev, non_blocking=True)
        self.seq_lens = torch.tensor(seq_lens, dtype=torch.int64).to(dev, non_blocking=True)
        prefix_lens_tensor = torch.tensor(prefix_lens, dtype=torch.int64).to(dev, non_blocking=True)
        extend_lens_tensor = torch.tensor(extend_lens, dtype=torch.int64).to(dev, non_blocking=True)

        for i, (req, seq_len, pre_len) in enumerate(zip(reqs, seq_lens, prefix_lens)):
            req.req_pool_idx = req_pool_indices[i]
            assert seq_len - pre_len == req.extend_input_len
            if pre_len > 0:
                pre = req.prefix_indices
                if not torch.is_tensor(pre):
                    pre = torch.tensor(pre, dtype=torch.int32)
                self.req_to_token_pool.write((req.req_pool_idx, slice(0, pre_len)),
                                             pre.to(device=dev, dtype=torch.int32))
            req.cached_tokens += pre_len - req.already_computed
            req.already_computed = seq_len
            # the request's logprob_start_len relative to this batch's new tokens (schedule_batch.py:1175-1183)
            if req.logprob_start_len >= pre_len:
                req.extend_logprob_start_len = min(req.logprob_start_len - pre_len, req.extend_input_len, req.seqlen - 1)
            else:
                req.extend_logprob_start_len = 0
        self.top_logprobs_nums = [r.top_logprobs_num for r in reqs]
        self.extend_logprob_start_lens = [r.extend_logprob_start_len for r in reqs]
        self.extend_input_logprob_token_ids = None
        if self.return_logprob:
            # the token whose log-probability each kept row reports = the NEXT prompt token; rows past the
            # prompt's end (the row that samples) are padded with 0 (schedule_batch.py:1185-1226)
            ids = []
            for req in reqs:
                start = max(len(req.prefix_indices), req.logprob_start_len)
                toks = req.origin_input_ids[start + 1: len(req.fill_ids) + 1]
                ids.extend(toks)
                ids.extend([0] * (req.extend_input_len - req.extend_logprob_start_len - len(toks)))
            self.extend_input_logprob_token_ids = torch.tensor(ids, dtype=torch.int64)

        if self.token_to_kv_pool_allocator.page_size == 1:
            self.out_cache_loc = self.alloc_token_slots(extend_num_tokens)
        else:
            # page-aligned slots continue each request's last page (schedule_batch.py alloc_paged_token_slots_extend;
            # last_loc = the slot of the last prefix token, -1 without a prefix: get_last_loc)
            last_loc = torch.tensor(
                [int(r.prefix_indices[-1]) if len(r.prefix_indices) > 0 else -1 for r in reqs], dtype=torch.int64).to(dev)
            self.out_cache_loc = self.token_to_kv_pool_allocator.alloc_extend(
                prefix_lens_tensor, self.seq_lens, last_loc, extend_num_tokens,
                prefix_lens_cpu=prefix_lens, seq_lens_cpu=seq_lens)
            if self.out_cache_loc is None:
                raise RuntimeError(f"Out of KV pages: need {extend_num_tokens} tokens, "
                                   f"available {self.token_to_kv_pool_allocator.available_size()}")
        self.seq_lens_cpu_list = list(seq_lens)
        self.seq_lens_sum = sum(seq_lens)
        self.extend_num_tokens = extend_num_tokens
        self.prefix_lens = prefix_lens
        self.extend_lens = extend_lens
        self.positions = None

        get_ops().write_req_to_token(self.req_to_token_pool.req_to_token, self.req_pool_indices,
                                     prefix_lens_tensor, self.seq_lens, extend_lens_tensor,
                                     self.out_cache_loc)
        self.sampling_info = SamplingBatchInfo.from_schedule_batch(self, self.model_config.vocab_size)

    def prepare_for_decode(self):
        self.forward_mode = ForwardMode.DECODE
        bs = len(self.reqs)
        self.input_ids = self.output_ids.to(torch.int64) if self.output_ids.dtype != torch.int64 else self.output_ids
        self.output_ids = None
        if self.token_to_kv_pool_allocator.page_size == 1:
            self.out_cache_loc = self.alloc_token_slots(bs)
        else:
            # the slot after each request's last token, or a fresh page (alloc_paged_token_slots_decode)
            rows = self.req_to_token_pool.req_to_token[self.req_pool_indices, self.seq_lens - 1]
            self.out_cache_loc = self.token_to_kv_pool_allocator.alloc_decode(
                self.seq_lens + 1, rows.to(torch.int64).contiguous(),
                seq_lens_cpu=[x + 1 for x in self.seq_lens_cpu_list])
            if self.out_cache_loc is None:
                raise RuntimeError(f"Out of KV pages for a decode step of {bs} requests")
        if self.positions is None or self.positions.numel() != bs:
            self.positions = torch.empty(bs, dtype=torch.int64, device=self.device)
        # seq_lens += 1; positions = seq_lens - 1; req_to_token[req, seq_lens - 1] = out_cache_loc
        get_ops().prepare_decode(self.req_to_token_pool.req_to_token, self.req_pool_indices,
                                 self.seq_lens, self.out_cache_loc, self.positions)
        self.seq_lens_cpu_list = [x + 1 for x in self.seq_lens_cpu_list]
        self.seq_lens_sum += bs

    def mix_with_running(self, running_batch: "ScheduleBatch"):
        """Append the running decode requests (already through prepare_for_decode) to this prefill
        batch as extends of length 1 over their own KV as prefix (schedule_batch.py:1310-1338)."""
        self.forward_mode = ForwardMode.MIXED
        running_bs = running_batch.batch_size()
        for req in running_batch.reqs:
            req.fill_ids = req.origin_input_ids + req.output_ids
            req.extend_input_len = 1
        input_ids = torch.cat([self.input_ids, running_batch.input_ids])
        out_cache_loc = torch.cat([self.out_cache_loc, running_batch.out_cache_loc])
        decode_prefix = [n - 1 for n in running_batch.seq_lens_cpu_list]   # KV before this step's token
        self.merge_batch(running_batch)
        self.input_ids = input_ids
        self.out_cache_loc = out_cache_loc
        self.prefix_lens = list(self.prefix_lens) + decode_prefix
        self.extend_lens = list(self.extend_lens) + [1] * running_bs
        self.extend_num_tokens += running_bs
        self.positions = None
        # the riders report no prompt log-probabilities (schedule_batch.py:1338)
        if self.extend_logprob_start_lens is not None:
            self.extend_logprob_start_lens = list(self.extend_logprob_start_lens) + [0] * running_bs
        if self.top_logprobs_nums is not None:
            self.top_logprobs_nums = list(self.top_logprobs_nums) + [0] * running_bs

    def get_model_worker_batch(self) -> ModelWorkerBatch:
        if self.forward_mode.is_decode_or_idle():
            extend_seq_lens = extend_prefix_lens = None
        else:
            extend_seq_lens, extend_prefix_lens = self.extend_lens, self.prefix_lens
        return ModelWorkerBatch(
            forward_mode=self.forward_mode,
            input_ids=self.input_ids,
            req_pool_indices=self.req_pool_indices,
            seq_lens=self.seq_lens,
            out_cache_loc=self.out_cache_loc,
            seq_lens_sum=self.seq_lens_sum,
            seq_lens_cpu=torch.tensor(self.seq_lens_cpu_list, dtype=torch.int64),
            return_logprob=self.return_logprob,
            extend_num_tokens=self.extend_num_tokens,
            extend_seq_lens=extend_seq_lens,
            extend_prefix_lens=extend_prefix_lens,
            sampling_info=self.sampling_info,
            positions=self.positions if self.forward_mode.is_decode() else None,
            top_logprobs_nums=self.top_logprobs_nums,
            extend_logprob_start_lens=None if self.forward_mode.is_decode_or_idle() else self.extend_logprob_start_lens,
            extend_input_logprob_token_ids=self.extend_input_logprob_token_ids,
        )

    def is_empty(self) -> bool:
        return len(self.reqs) == 0

    def filter_batch(self, keep_indices: Optional[List[int]] = None, chunked_req_to_exclude=None):
        """Drop finished (and still-chunked) requests from a decode batch (schedule_batch.py:1575-1622)."""
        if keep_indices is None:
            keep_indices = [i for i, r in enumerate(self.reqs) if not r.finished and r is not chunked_req_to_exclude]
        if len(keep_indices) == len(self.reqs):
            return
        if not keep_indices:
            self.reqs = []
            self.seq_lens_cpu_list = []
            self.seq_lens_sum = 0
            self.req_pool_indices = self.seq_lens = self.output_ids = self.positions = None
            return
        keep = torch.tensor(keep_indices, dtype=torch.int64).to(self.device, non_blocking=True)
        self.reqs = [self.reqs[i] for i in keep_indices]
        self.req_pool_indices = self.req_pool_indices[keep]
        self.seq_lens = self.seq_lens[keep]
        self.seq_lens_cpu_list = [self.seq_lens_cpu_list[i] for i in keep_indices]
        self.seq_lens_sum = sum(self.seq_lens_cpu_list)
        if self.output_ids is not None:
            self.output_ids = self.output_ids[keep]
        self.positions = None
        self.out_cache_loc = None
        self.sampling_info.filter_batch(keep_indices, keep)

    def merge_batch(self, other: "ScheduleBatch"):
        """Append a just-prefilled batch to the running decode batch (schedule_batch.py:1624-1668)."""
        if self.is_empty():
            self.__dict__.update(other.__dict__)
            return
        self.sampling_info.merge_batch(other.sampling_info)
        self.req_pool_indices = torch.cat([self.req_pool_indices, other.req_pool_indices])
        self.seq_lens = torch.cat([self.seq_lens, other.seq_lens])
        self.seq_lens_cpu_list = self.seq_lens_cpu_list + other.seq_lens_cpu_list
        self.seq_lens_sum += other.seq_lens_sum
        self.out_cache_loc = None
        self.positions = None
        if self.output_ids is not None and other.output_ids is not None:
            self.output_ids = torch.cat([self.output_ids.to(torch.int64), other.output_ids.to(torch.int64)])
        self.reqs = self.reqs + other.reqs
        self.return_logprob = self.return_logprob or other.return_logprob

    def release(self):
        """Return every slot of the batch (what the scheduler's cache_finished_req does)."""
        for r, n in zip(self.reqs, self.seq_lens_cpu_list):
            if r.req_pool_idx is None:
                continue
            slots = self.req_to_token_pool.req_to_token[r.req_pool_idx, :n].to(torch.int64)
            self.token_to_kv_pool_allocator.free(slots)
            self.req_to_token_pool.free(r.req_pool_idx)
            r.req_pool_idx = None
