"""HipGraphRunner: capture the decode step per batch size into a HIP graph and replay it from
static input buffers -- the role of the reference's CudaGraphRunner
(python/sglang/srt/model_executor/cuda_graph_runner.py: can_run :342, capture_one_batch_size
:456, replay_prepare :626-700, replay :702-730).  ~330 kernel launches per step collapse into
one graph launch; padded rows use seq_len fill value 1 and KV slot 0 (the sink row)."""
import bisect
from typing import Dict, List

import torch

from ..layers.logits_processor import LogitsProcessorOutput
from ..ops_backend import get_ops
from .forward_batch_info import ForwardBatch, ForwardMode


def get_batch_sizes_to_capture(max_bs: int) -> List[int]:
    sizes = [1, 2, 4, 8, 12, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256]
    out = [b for b in sizes if b <= max_bs]
    if max_bs not in out:
        out.append(max_bs)
    return sorted(out)


class HipGraphRunner:
    def __init__(self, model_runner):
        self.model_runner = model_runner
        sa = model_runner.server_args
        self.capture_bs = sorted(sa.cuda_graph_bs) if sa.cuda_graph_bs else get_batch_sizes_to_capture(sa.cuda_graph_max_bs)
        self.max_bs = max(self.capture_bs)
        dev = model_runner.device
        self.seq_len_fill_value = model_runner.attn_backend.get_cuda_graph_seq_len_fill_value()
        # static inputs (cuda_graph_runner.py:270-290)
        self.input_ids = torch.zeros(self.max_bs, dtype=torch.int64, device=dev)
        self.req_pool_indices = torch.zeros(self.max_bs, dtype=torch.int64, device=dev)
        self.seq_lens = torch.full((self.max_bs,), self.seq_len_fill_value, dtype=torch.int64, device=dev)
        self.out_cache_loc = torch.zeros(self.max_bs, dtype=torch.int64, device=dev)
        self.positions = torch.zeros(self.max_bs, dtype=torch.int64, device=dev)
        self.graphs: Dict[int, torch.cuda.CUDAGraph] = {}
        self.outputs: Dict[int, torch.Tensor] = {}
        self.pool = None
        model_runner.attn_backend.init_cuda_graph_state(self.max_bs, self.max_bs)
        self.capture()

    def can_run(self, forward_batch: ForwardBatch) -> bool:
        return forward_batch.forward_mode.is_decode() and forward_batch.batch_size <= self.max_bs

    def _static_batch(self, bs: int) -> ForwardBatch:
        mr = self.model_runner
        return ForwardBatch(
            forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=self.input_ids[:bs],
            req_pool_indices=self.req_pool_indices[:bs], seq_lens=self.seq_lens[:bs],
            out_cache_loc=self.out_cache_loc[:bs], seq_lens_sum=bs * self.seq_len_fill_value,
            positions=self.positions[:bs], req_to_token_pool=mr.req_to_token_pool,
            token_to_kv_pool=mr.token_to_kv_pool, attn_backend=mr.attn_backend)

    def capture(self):
        from ..distributed.parallel_state import check_collectives, tp_barrier

        mr = self.model_runner
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for bs in reversed(self.capture_bs):  # largest first: it sizes the shared pool
                fb = self._static_batch(bs)
                mr.attn_backend.init_forward_metadata_capture_cuda_graph(
                    bs, bs, fb.req_pool_indices, fb.seq_lens, None, ForwardMode.DECODE, None)
                run = lambda: mr.model.forward(fb.input_ids, fb.positions, fb).next_token_logits
                # ranks reach this point at different times (one may still be compiling or paging in the
                # library); line them up on the host so that the first collective of the warm-up runs does
                # not spend its bounded wait on that skew
                torch.cuda.current_stream().synchronize()
                tp_barrier()
                for _ in range(2):  # warm-up: grow every workspace before capture
                    run()
                torch.cuda.current_stream().synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool, stream=stream):
                    out = run()
                self.pool = self.pool or g.pool()
                self.graphs[bs] = g
                self.outputs[bs] = out
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        # ranks capture at different speeds: a collective of the warm-up runs that gave up waiting
        # for a slow peer must fail here, not as wrong tokens later
        check_collectives()

    def replay(self, forward_batch: ForwardBatch) -> LogitsProcessorOutput:
        raw_bs = forward_batch.batch_size
        bs = self.capture_bs[bisect.bisect_left(self.capture_bs, raw_bs)]
        if bs != raw_bs:
            self.seq_lens.fill_(self.seq_len_fill_value)
            self.out_cache_loc.zero_()
            self.req_pool_indices.zero_()
        dsts = [self.input_ids, self.req_pool_indices, self.seq_lens, self.out_cache_loc, self.positions]
        srcs = [forward_batch.input_ids, forward_batch.req_pool_indices, forward_batch.seq_lens,
                forward_batch.out_cache_loc, forward_batch.positions]
        ops = get_ops()
        if all(t.dtype == torch.int64 and t.is_contiguous() for t in srcs):
            ops.copy_decode_inputs(dsts, srcs, raw_bs)   # one launch instead of five copies
        else:
            for d, s_ in zip(dsts, srcs):
                d[:raw_bs].copy_(s_)
        self.model_runner.attn_backend.init_forward_metadata_replay_cuda_graph(
            bs, self.req_pool_indices, self.seq_lens, forward_batch.seq_lens_sum, None,
            ForwardMode.DECODE, None, forward_batch.seq_lens_cpu)
        self.graphs[bs].replay()
        return LogitsProcessorOutput(next_token_logits=self.outputs[bs][:raw_bs])
