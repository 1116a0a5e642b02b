"""Continuous-batching loop that drives the hot path: the part of the reference's Scheduler that
decides what runs next and folds results back into the batches
(python/sglang/srt/managers/scheduler.py: get_next_batch_to_run :1608-1680, get_new_batch_prefill
:1682-1830, update_running_batch :1832-1866, run_batch :1868-1930;
schedule_policy.py PrefillAdder: add_one_req / add_chunked_req budgets;
scheduler_output_processor_mixin.py: process_batch_result_prefill / _decode).

Kept: FCFS admission under three budgets (free KV slots incl. the tokens running requests may
still generate, max_prefill_tokens per prefill batch, chunked_prefill_size per chunk), chunked
prefill of long prompts (the unfinished request keeps its KV slots and re-enters the next prefill
batch with them as its prefix), prefill-first policy, merge of prefilled requests into the running
decode batch, per-step retirement of finished requests with slot release.
Optional radix prefix cache (mem_cache/radix_cache.py): new requests start from the longest cached
prefix (its slots become prefix_indices of the extend), finished and chunk-boundary requests hand
their KV to the tree, and tree leaves are evicted LRU-first when the allocator runs short.
Retraction under memory pressure (update_running_batch :1828-1866, ScheduleBatch.retract_decode
schedule_batch.py:1362-1484): admission reserves only new_token_ratio of what the running requests
may still generate; when a decode step finds the pool short, the requests with the fewest outputs
(longest prompts first among equals) give their slots back and return to the waiting queue with
their outputs so far, to be prefilled again over prompt + outputs, and the ratio is re-estimated.
Mixed batches (enable_mixed_chunk): the running decode requests join a prefill batch as extends of
length 1 (ScheduleBatch.mix_with_running, schedule_batch.py:1310-1338).
Left out on purpose: priorities, streaming, the IPC around it."""
from typing import Dict, List, Optional

import torch

from ..distributed.parallel_state import check_collectives
from ..mem_cache.radix_cache import RadixCache
from ..model_executor.forward_batch_info import ForwardBatch
from .schedule_batch import Req, ScheduleBatch


# global_config.py of the reference: retract_decode_steps, default_init_new_token_ratio,
# default_min_new_token_ratio_factor, default_new_token_ratio_decay_steps
RETRACT_DECODE_STEPS = 20
DEFAULT_INIT_NEW_TOKEN_RATIO = 0.7
MIN_NEW_TOKEN_RATIO_FACTOR = 0.14
NEW_TOKEN_RATIO_DECAY_STEPS = 600
# schedule_policy.py CLIP_MAX_NEW_TOKENS_ESTIMATION
CLIP_MAX_NEW_TOKENS = 4096


class Scheduler:
    def __init__(self, model_runner, eos_token_id: Optional[int] = None, enable_radix_cache: bool = False):
        self.mr = model_runner
        self.page_size = getattr(model_runner.token_to_kv_pool_allocator, "page_size", 1)
        self.tree_cache = RadixCache(model_runner.req_to_token_pool, model_runner.token_to_kv_pool_allocator,
                                     disable=not enable_radix_cache, page_size=self.page_size)
        self.cached_prefix_tokens = 0
        sa = model_runner.server_args
        self.eos_token_id = eos_token_id
        self.max_running_requests = model_runner.req_to_token_pool.size
        cps = sa.chunked_prefill_size
        self.chunked_prefill_size = cps if cps is not None and cps > 0 else None
        self.max_prefill_tokens = sa.max_prefill_tokens
        self.waiting_queue: List[Req] = []
        self.running_batch: ScheduleBatch = self._new_batch([])
        self.chunked_req: Optional[Req] = None
        self.finished: Dict[object, Req] = {}
        self.num_prefill_batches = 0
        self.num_decode_steps = 0
        self.evicted_tokens = 0
        # scheduler.py:478-491
        self.init_new_token_ratio = min(DEFAULT_INIT_NEW_TOKEN_RATIO * getattr(sa, "schedule_conservativeness", 1.0), 1.0)
        self.min_new_token_ratio = min(self.init_new_token_ratio * MIN_NEW_TOKEN_RATIO_FACTOR, 1.0)
        self.new_token_ratio_decay = (self.init_new_token_ratio - self.min_new_token_ratio) / NEW_TOKEN_RATIO_DECAY_STEPS
        self.new_token_ratio = self.init_new_token_ratio
        self.total_retracted_reqs = 0
        self.enable_mixed_chunk = bool(getattr(sa, "enable_mixed_chunk", False))
        self.num_mixed_batches = 0

    # ---- queue ----
    def add_request(self, req: Req) -> None:
        total = len(req.origin_input_ids) + req.sampling_params.max_new_tokens
        if total > self.mr.req_to_token_pool.max_context_len:
            raise ValueError(f"request {req.rid}: {total} tokens exceed the context length")
        self.waiting_queue.append(req)

    def _new_batch(self, reqs) -> ScheduleBatch:
        mr = self.mr
        return ScheduleBatch.init_new(reqs=reqs, req_to_token_pool=mr.req_to_token_pool,
                                      token_to_kv_pool_allocator=mr.token_to_kv_pool_allocator,
                                      tree_cache=None, model_config=mr.model_config)

    # ---- admission (PrefillAdder) ----
    def _reserved_decode_tokens(self) -> int:
        """Slots the running requests are expected to still need: the estimated fraction
        new_token_ratio of what they may generate (PrefillAdder.rem_total_tokens, schedule_policy.py)."""
        return int(sum(min(r.sampling_params.max_new_tokens - len(r.output_ids), CLIP_MAX_NEW_TOKENS) * self.new_token_ratio
                       + (self.page_size if self.page_size > 1 else 0) for r in self.running_batch.reqs))

    def get_new_batch_prefill(self) -> Optional[ScheduleBatch]:
        if not self.waiting_queue and self.chunked_req is None:
            return None
        alloc = self.mr.token_to_kv_pool_allocator
        rem_total = alloc.available_size() + self.tree_cache.evictable_size() - self._reserved_decode_tokens()
        rem_input = self.max_prefill_tokens
        rem_chunk = self.chunked_prefill_size
        if self.enable_mixed_chunk:
            # the decode tokens that will ride along count against every budget (PrefillAdder
            # mixed_with_decode_tokens, schedule_policy.py)
            mixed = len(self.running_batch.reqs)
            rem_total -= mixed * self.page_size
            rem_input -= mixed
            if rem_chunk is not None:
                rem_chunk -= mixed
        free_reqs = self.mr.req_to_token_pool.available_size()
        can_run: List[Req] = []

        def take(req: Req, prefix) -> bool:
            """Admit req (whole, or its next chunk); returns False when the batch is full."""
            nonlocal rem_total, rem_input, rem_chunk
            matched_node = None
            if prefix is None and req.last_node is None:
                # longest cached prefix, always leaving one token to compute (schedule_batch.py
                # Req.init_next_round_input / adjust_max_prefix_ids)
                key = req.origin_input_ids + req.output_ids
                prefix, matched_node = self.tree_cache.match_prefix(key[:-1])
            req.init_next_round_input(prefix)
            locked = 0
            if matched_node is not None:
                before = self.tree_cache.evictable_size()
                self.tree_cache.inc_lock_ref(matched_node)   # its slots leave the evictable pool
                locked = before - self.tree_cache.evictable_size()
                rem_total -= locked

            def refuse() -> bool:
                nonlocal rem_total
                if matched_node is not None:
                    self.tree_cache.dec_lock_ref(matched_node)
                    rem_total += locked
                req.init_next_round_input(None)
                return False

            need_all = req.extend_input_len + req.sampling_params.max_new_tokens - len(req.output_ids)
            if self.page_size > 1:   # a request occupies whole pages (its last one partly)
                need_all = (need_all + self.page_size - 1) // self.page_size * self.page_size + self.page_size
            if need_all > rem_total:
                return refuse()
            n = req.extend_input_len
            truncate = rem_chunk is not None and n > rem_chunk
            # chunks end on page boundaries (PrefillAdder trunc_len, schedule_policy.py)
            chunk = rem_chunk // self.page_size * self.page_size if truncate else 0
            if (truncate and chunk <= 0) or (not truncate and n > rem_input and can_run):
                return refuse()
            if matched_node is not None:
                req.last_node = matched_node
                self.cached_prefix_tokens += len(prefix)
            if truncate:
                # truncate to the chunk budget (add_chunked_req / add_one_req trunc_len)
                n = chunk
                req.fill_ids = req.fill_ids[: len(req.prefix_indices) + n]
                req.extend_input_len = n
                req.is_chunked += 1
                self.chunked_req = req
            can_run.append(req)
            rem_total -= need_all
            rem_input -= n
            if rem_chunk is not None:
                rem_chunk -= n
            return True

        if self.chunked_req is not None:
            # the unfinished prompt goes first, carrying the slots it already owns as its prefix
            req = self.chunked_req
            self.chunked_req = None
            done = req.already_computed
            prefix = self.mr.req_to_token_pool.req_to_token[req.req_pool_idx, :done].clone()
            self.mr.req_to_token_pool.free(req.req_pool_idx)   # prepare_for_extend allocates a row again
            req.req_pool_idx = None
            free_reqs += 1
            rem_total += 0   # its slots are already allocated and stay allocated
            if not take(req, prefix):
                raise RuntimeError("no room to continue a chunked prefill")
        while self.waiting_queue and self.chunked_req is None:
            if len(can_run) >= free_reqs or len(can_run) + len(self.running_batch.reqs) >= self.max_running_requests:
                break
            if rem_input <= 0 or (rem_chunk is not None and rem_chunk <= 0):
                break
            if not take(self.waiting_queue[0], None):
                break
            self.waiting_queue.pop(0)
        if not can_run:
            return None
        self._make_room(sum(r.extend_input_len for r in can_run))
        batch = self._new_batch(can_run)
        batch.prepare_for_extend()
        # mixed chunk (scheduler.py get_new_batch_prefill, is_mixed_chunk branch): the running
        # requests take their decode step inside this forward instead of waiting for it
        if self.enable_mixed_chunk and not self.running_batch.is_empty() and self.check_decode_mem(self.running_batch):
            self.running_batch.prepare_for_decode()
            batch.mix_with_running(self.running_batch)
            self.running_batch = self._new_batch([])
            self.num_mixed_batches += 1
        return batch

    def _make_room(self, need: int) -> None:
        """Evict least-recently-used cached prefixes when the allocator alone cannot serve `need`."""
        short = need - self.mr.token_to_kv_pool_allocator.available_size()
        if short > 0:
            self.evicted_tokens += self.tree_cache.evict(short)

    # ---- one iteration of the event loop ----
    def get_next_batch_to_run(self) -> Optional[ScheduleBatch]:
        new_batch = self.get_new_batch_prefill()
        if new_batch is not None:
            return new_batch
        if self.running_batch.is_empty():
            return None
        return self.update_running_batch(self.running_batch)

    # ---- decode under memory pressure ----
    def _decode_slots_needed(self, batch: ScheduleBatch) -> int:
        """Slots the next decode step allocates (new_page_count_next_decode * page_size)."""
        if self.page_size == 1:
            return len(batch.reqs)
        return sum(1 for n in batch.seq_lens_cpu_list if n % self.page_size == 0) * self.page_size

    def check_decode_mem(self, batch: ScheduleBatch) -> bool:
        need = self._decode_slots_needed(batch)
        self._make_room(need)
        return self.mr.token_to_kv_pool_allocator.available_size() >= need

    def retract_decode(self, batch: ScheduleBatch) -> List[Req]:
        alloc, pool = self.mr.token_to_kv_pool_allocator, self.mr.req_to_token_pool
        reqs = batch.reqs
        # popped from the end: fewest outputs first, longest prompt first among equals
        order = sorted(range(len(reqs)), key=lambda i: (len(reqs[i].output_ids), -len(reqs[i].origin_input_ids)),
                       reverse=True)
        per_req = RETRACT_DECODE_STEPS + (self.page_size if self.page_size > 1 else 0)
        retracted: List[Req] = []
        first = True
        while alloc.available_size() < len(order) * per_req or first:
            if len(order) == 1:
                if alloc.available_size() < self._decode_slots_needed(batch) and not retracted:
                    raise RuntimeError("No space left for only one request")
                break
            first = False
            idx = order.pop()
            req = reqs[idx]
            retracted.append(req)
            # what the prefix tree holds stays there (unlocked); the rest goes back to the allocator
            own_from = 0 if self.tree_cache.disable else len(req.prefix_indices) // self.page_size * self.page_size
            slots = pool.req_to_token[req.req_pool_idx, own_from:batch.seq_lens_cpu_list[idx]].to(torch.int64)
            alloc.free(slots)
            pool.free(req.req_pool_idx)
            if req.last_node is not None:
                self.tree_cache.dec_lock_ref(req.last_node)
            self._make_room(len(order) * per_req)
            req.reset_for_retract()
        batch.filter_batch(keep_indices=sorted(order))
        decoded = sum(len(r.output_ids) for r in batch.reqs)
        budget = sum(r.sampling_params.max_new_tokens for r in batch.reqs)
        self.new_token_ratio = min(1.0, (decoded + RETRACT_DECODE_STEPS * len(batch.reqs)) / max(budget, 1))
        return retracted

    def update_running_batch(self, batch: ScheduleBatch) -> ScheduleBatch:
        if not self.check_decode_mem(batch):
            retracted = self.retract_decode(batch)
            self.waiting_queue.extend(retracted)
            self.total_retracted_reqs += len(retracted)
        else:
            self.new_token_ratio = max(self.new_token_ratio - self.new_token_ratio_decay, self.min_new_token_ratio)
        batch.prepare_for_decode()
        return batch

    @torch.no_grad()
    def run_batch(self, batch: ScheduleBatch) -> torch.Tensor:
        fb = ForwardBatch.init_new(batch.get_model_worker_batch(), self.mr)
        logits_output, _ = self.mr.forward(fb)
        return self.mr.sample(logits_output, fb)

    def _check_finished(self, req: Req) -> None:
        if len(req.output_ids) >= req.sampling_params.max_new_tokens:
            req.finished = True
        elif (self.eos_token_id is not None and not req.sampling_params.ignore_eos
              and req.output_ids and req.output_ids[-1] == self.eos_token_id):
            req.finished = True

    def _release(self, req: Req, seq_len: int) -> None:
        if not self.tree_cache.disable:
            assert seq_len == len(req.origin_input_ids) + len(req.output_ids) - 1
            self.tree_cache.cache_finished_req(req)
        else:
            pool = self.mr.req_to_token_pool
            slots = pool.req_to_token[req.req_pool_idx, :seq_len].to(torch.int64)
            self.mr.token_to_kv_pool_allocator.free(slots)
            pool.free(req.req_pool_idx)
        req.req_pool_idx = None
        req.last_node = None
        self.finished[req.rid] = req

    def process_batch_result(self, batch: ScheduleBatch, next_token_ids: torch.Tensor) -> None:
        ids = next_token_ids.tolist()
        check_collectives()    # the read-back has synchronised: a collective that lost a peer fails the step here
        if batch.forward_mode.is_extend():
            self.num_prefill_batches += 1
            keep = []
            for i, (req, tok) in enumerate(zip(batch.reqs, ids)):
                if req.is_chunked > 0 and req is self.chunked_req:
                    # mid-prompt: the sampled token is meaningless (scheduler_output_processor_mixin);
                    # what is computed so far goes to the tree
                    self.tree_cache.cache_unfinished_req(req)
                    continue
                req.is_chunked = 0
                req.output_ids.append(tok)
                self._check_finished(req)
                if req.finished:
                    self._release(req, batch.seq_lens_cpu_list[i])
                else:
                    keep.append(i)
            batch.output_ids = next_token_ids.to(torch.int64)
            batch.filter_batch(keep_indices=keep)
            if not batch.is_empty():
                self.running_batch.merge_batch(batch)
        else:
            self.num_decode_steps += 1
            keep = []
            for i, (req, tok) in enumerate(zip(batch.reqs, ids)):
                req.output_ids.append(tok)
                self._check_finished(req)
                if req.finished:
                    self._release(req, batch.seq_lens_cpu_list[i])
                else:
                    keep.append(i)
            batch.output_ids = next_token_ids.to(torch.int64)
            batch.filter_batch(keep_indices=keep)

    def step(self) -> bool:
        batch = self.get_next_batch_to_run()
        if batch is None:
            self.new_token_ratio = self.init_new_token_ratio   # idle: forget the estimate (scheduler.py:813)
            return False
        self.process_batch_result(batch, self.run_batch(batch))
        return True

    # ---- overlap mode: the next step is launched before the last one's tokens are read ----
    # (scheduler.py event_loop_overlap :749-789, tp_worker_overlap_thread.py:53-292.  The reference
    # needs a worker thread and a map of "future" token ids because its sampled tokens pass through
    # the host; here a step's token ids stay on the device and are the next step's input as they
    # are, so one thread that launches first and reads results one step late gives the same overlap.)
    def _launch(self, batch: ScheduleBatch):
        """Launch `batch`, start the copy of its token ids to the host, and bring the batches to the
        state the next launch needs, none of which depends on the ids' values."""
        ids = self.run_batch(batch)
        host = torch.empty(ids.shape, dtype=ids.dtype, device="cpu", pin_memory=ids.is_cuda)
        host.copy_(ids, non_blocking=True)
        done = None
        if ids.is_cuda:
            done = torch.cuda.Event()
            done.record()
        mid_prompt = self.chunked_req if batch.forward_mode.is_extend() else None
        launched = dict(reqs=list(batch.reqs), ids=host, done=done, extend=batch.forward_mode.is_extend(),
                        mid_prompt=mid_prompt)
        batch.output_ids = ids.to(torch.int64)
        if batch.forward_mode.is_extend():
            self.num_prefill_batches += 1
            keep = []
            for i, req in enumerate(batch.reqs):
                if req.is_chunked > 0 and req is mid_prompt:
                    self.tree_cache.cache_unfinished_req(req)
                    continue
                req.is_chunked = 0
                keep.append(i)
            batch.filter_batch(keep_indices=keep)
            if not batch.is_empty():
                self.running_batch.merge_batch(batch)
        else:
            self.num_decode_steps += 1
        return launched

    def _release_late(self, req: Req) -> None:
        """A request whose end became known one step late: it may already sit in a launched decode
        step, whose slot goes back with the rest."""
        if req.req_pool_idx is None:            # retracted meanwhile: it holds nothing
            self.waiting_queue = [r for r in self.waiting_queue if r is not req]
            self.finished[req.rid] = req
            return
        rb = self.running_batch
        i = next(k for k, r in enumerate(rb.reqs) if r is req)
        held = rb.seq_lens_cpu_list[i]
        cached = len(req.origin_input_ids) + len(req.output_ids) - 1
        if not self.tree_cache.disable:
            self.tree_cache.cache_finished_req(req, num_extra_slots=held - cached)
        else:
            pool = self.mr.req_to_token_pool
            self.mr.token_to_kv_pool_allocator.free(pool.req_to_token[req.req_pool_idx, :held].to(torch.int64))
            pool.free(req.req_pool_idx)
        req.req_pool_idx = None
        req.last_node = None
        self.finished[req.rid] = req

    def _collect(self, launched) -> None:
        if launched["done"] is not None:
            launched["done"].synchronize()
        check_collectives(wait=False)   # this step is complete; the next one is in flight and is not waited for
        for req, tok in zip(launched["reqs"], launched["ids"].tolist()):
            if req is launched["mid_prompt"] or req.finished:
                continue                        # mid-prompt sample, or a step past the request's end
            req.output_ids.append(tok)
            self._check_finished(req)
            if req.finished:
                self._release_late(req)

    def run_until_done_overlap(self, max_iters: int = 1 << 30, arrivals=None) -> Dict[object, Req]:
        """arrivals: {iteration: [requests that arrive before it]} (a stand-in for the request socket)."""
        assert not self.enable_mixed_chunk, "overlap mode and mixed batches are not combined"
        arrivals = dict(arrivals or {})
        in_flight = None
        it = 0
        while (self.waiting_queue or self.chunked_req is not None or not self.running_batch.is_empty()
               or in_flight is not None or arrivals) and it < max_iters:
            for req in arrivals.pop(it, []):
                self.add_request(req)
            rb = self.running_batch
            if any(r.finished for r in rb.reqs):
                rb.filter_batch(keep_indices=[i for i, r in enumerate(rb.reqs) if not r.finished])
            batch = self.get_next_batch_to_run()
            launched = self._launch(batch) if batch is not None else None
            if in_flight is not None:
                self._collect(in_flight)        # the host reads step N while the device runs step N+1
            elif launched is None and not arrivals:
                raise RuntimeError("scheduler stalled: a waiting request does not fit the KV pool")
            in_flight = launched
            it += 1
        return self.finished

    def run_until_done(self, max_iters: int = 1 << 30) -> Dict[object, Req]:
        it = 0
        while (self.waiting_queue or self.chunked_req is not None or not self.running_batch.is_empty()) and it < max_iters:
            if not self.step():
                raise RuntimeError("scheduler stalled: a waiting request does not fit the KV pool")
            it += 1
        return self.finished
