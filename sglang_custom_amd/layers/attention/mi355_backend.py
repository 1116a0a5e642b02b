"""Mi355AttnBackend: hand-written HIP attention behind the AttentionBackend surface.

Plays the role of TritonAttnBackend (python/sglang/srt/layers/attention/triton_backend.py:
init_forward_metadata :160-336, graph variants :338-627, forward_extend :632-685,
forward_decode :687-732) with two deliberate differences:
  * decode reads the req_to_token page table directly (as the reference's CPU kernel does,
    sgl-kernel/csrc/cpu/decode.cpp:913-916), so no kv_indices gather and no per-forward
    allocation or device->host read is needed: metadata is a split count chosen on the host;
  * the per-forward scratch (split partials) lives in grow-only buffers owned by the backend
    (outgrown buffers are kept alive: captured graphs hold their addresses).
Sliding-window layers follow the reference's scheme (triton_backend.py:188-205, 302-310, 656-663,
711-716, 927-973): the keys handed to the kernels are the LAST min(len, window + 1) cached tokens,
gathered into window_kv_indptr / window_kv_indices (prefix tokens for an extend, all tokens for a
decode); the extend kernel additionally masks the prefix phase with q_index <= k_index + window.
Called once per layer; the backend owns the KV write when save_kv_cache is set.
"""
import dataclasses
import os
from typing import Optional

import torch

from ...ops_backend import get_ops
from .base_attn_backend import AttentionBackend

MAX_KV_SPLITS = 64
# MLA extend over cached rows: the kernel in which four tokens share a pass over the rows ("tiled"), or one
# decode-style item per new token ("per_token").  The tiled kernel is launched with a quarter of the
# workgroups, so batches too small to fill the chip with it stay on the per-token pass.
MLA_EXTEND_TILED = os.environ.get("MI355_MLA_EXTEND", "tiled") == "tiled"
MLA_EXTEND_TILED_MIN_WGS = 256
# MLA decode with RoPE and the latent-row write inside the attention launch ("on"), or rope / cache write /
# attention as separate launches ("off"); the reference's switch is SGLANG_ROCM_FUSED_DECODE_MLA.
MLA_FUSED_ROPE = os.environ.get("MI355_MLA_FUSED_ROPE", "on") == "on"


@dataclasses.dataclass
class ForwardMetadata:
    num_kv_splits: int = 1
    tickets: Optional[torch.Tensor] = None   # arrival counters of the in-kernel split merge (zero between launches)
    kv_chunk_keys: Optional[torch.Tensor] = None   # int32 [1] on the device: the batch's keys per KV split (per-sequence split plan)
    attn_logits: Optional[torch.Tensor] = None
    attn_lse: Optional[torch.Tensor] = None
    qo_indptr: Optional[torch.Tensor] = None
    kv_indptr: Optional[torch.Tensor] = None
    kv_indices: Optional[torch.Tensor] = None
    max_extend_len: int = 0
    window_kv_indptr: Optional[torch.Tensor] = None
    window_kv_indices: Optional[torch.Tensor] = None
    window_num_kv_splits: int = 1


class Mi355AttnBackend(AttentionBackend):
    def __init__(self, model_runner):
        super().__init__()
        self.device = model_runner.device
        self.req_to_token = model_runner.req_to_token_pool.req_to_token
        cfg = model_runner.model_config
        tp = model_runner.tp_size
        self.num_head = cfg.num_attention_heads // tp
        self.num_kv_head = max(1, cfg.num_key_value_heads // tp)
        self.is_mla = hasattr(model_runner.token_to_kv_pool, "kv_lora_rank")
        self.kv_fp8 = getattr(model_runner.token_to_kv_pool, "dtype", None) == torch.float8_e4m3fn
        if self.is_mla:
            # what the split plan counts is workgroups per sequence, one per 16 query heads
            self.num_kv_head = (self.num_head + 15) // 16
        # value width as the pool stores it (an MLA pool's value is the first kv_lora_rank features of
        # its rows); read from the buffer so that any pool with the reference's accessors works
        self.v_head_dim = model_runner.token_to_kv_pool.get_value_buffer(0).shape[-1]
        self.max_context_len = int(getattr(cfg, "context_len", 0) or self.req_to_token.shape[1])
        sw = getattr(model_runner, "sliding_window_size", None)
        self.sliding_window_size = sw if sw is not None and sw > 0 else None
        self.forward_metadata: Optional[ForwardMetadata] = None
        self._logits = None
        self._lse = None
        self._tickets = None
        self._retired = []      # outgrown scratch: captured graphs may still point into it
        self._retired_tickets = []   # outgrown counter arrays, for the same reason; cleared with the live one
        self._dummy_idx = torch.zeros(1, dtype=torch.int32, device=self.device)
        # keys per KV split of the current batch, read by the decode kernels when they run (also inside
        # replayed graphs): a sequence shorter than the longest one is cut into fewer splits
        # (get_num_kv_splits_triton of the reference, triton_backend.py:875-924)
        self._chunk_keys = torch.ones(1, dtype=torch.int32, device=self.device)
        self.graph_num_splits = {}
        self.graph_window_indptr = None
        self.graph_window_indices = None

    # ---- scratch ----
    def _scratch(self, bs: int, splits: int):
        need = bs * self.num_head * splits
        if self._lse is None or self._lse.numel() < need:
            if self._lse is not None:
                self._retired.append((self._logits, self._lse))
            self._logits = torch.empty(need * self.v_head_dim, dtype=torch.float32, device=self.device)
            self._lse = torch.empty(need, dtype=torch.float32, device=self.device)
        if self._tickets is None or self._tickets.numel() < bs * self.num_head:
            # one counter per (sequence, head); the kernels hand them back zeroed.  An outgrown array
            # stays alive: captured graphs may still point into it.
            if self._tickets is not None:
                self._retired_tickets.append(self._tickets)
            self._tickets = torch.zeros(max(bs * self.num_head, 1024), dtype=torch.int32, device=self.device)
        return self._logits, self._lse

    def _window(self, lens: torch.Tensor, lens_cpu, req_pool_indices, bs: int, indptr=None, indices=None):
        """The last min(len, window + 1) tokens of every request as (indptr, indices, longest)."""
        w1 = self.sliding_window_size + 1
        wl = torch.clamp(lens, max=w1)
        if indptr is None:
            indptr = torch.zeros(bs + 1, dtype=torch.int32, device=self.device)
        indptr[1: bs + 1] = torch.cumsum(wl, dim=0)
        if indices is None:
            total = sum(min(int(n), w1) for n in lens_cpu) if lens_cpu is not None else int(indptr[bs].item())
            indices = torch.empty(max(total, 1), dtype=torch.int32, device=self.device)
        if bs > 0:
            get_ops().create_kv_indices(self.req_to_token, req_pool_indices, wl, indptr[: bs + 1],
                                        (lens - wl).to(torch.int32), indices)
        return indptr[: bs + 1], indices

    def _plan_splits(self, bs: int, max_seq_len: int) -> int:
        if self.is_mla:
            # the MLA kernel keeps two workgroups per CU busy and streams 1152-byte rows: it is fastest at
            # 512 workgroups (64 sequences x 16 heads x 4096 keys: 190 / 107 / 75 / 68 us at 1 / 2 / 4 / 8 splits,
            # 91 at 10) with at least 256 keys per split (64 x 1024 keys: 31.6 us at 4 splits, 34.5 at 8)
            # (the ring over an e4m3fn latent cache has the same shape: 64 x 16 x 4096: 103 / 68 / 57 / 81 us at
            # 2 / 4 / 8 / 16 splits)
            wgs = max(1, bs * self.num_kv_head)
            return max(1, min(MAX_KV_SPLITS, (512 + wgs - 1) // wgs, max(1, max_seq_len // 256)))
        return get_ops().decode_num_splits(bs, self.num_kv_head, max_seq_len, MAX_KV_SPLITS)

    # ---- eager ----
    def init_forward_metadata(self, forward_batch):
        ops = get_ops()
        bs = forward_batch.batch_size
        if forward_batch.forward_mode.is_decode_or_idle():
            if forward_batch.seq_lens_cpu is not None and bs > 0:
                max_len = int(forward_batch.seq_lens_cpu.max())
            else:
                max_len = max(1, forward_batch.seq_lens_sum // max(bs, 1))
            splits = self._plan_splits(bs, max_len)
            wsplits = self._plan_splits(bs, min(max_len, self.sliding_window_size + 1)) if self.sliding_window_size else 1
            logits, lse = self._scratch(bs, max(splits, wsplits)) if max(splits, wsplits) > 1 else (None, None)
            md = ForwardMetadata(num_kv_splits=splits, attn_logits=logits, attn_lse=lse, window_num_kv_splits=wsplits,
                                 tickets=self._tickets if logits is not None else None,
                                 kv_chunk_keys=self._chunk_keys if splits > 1 else None)
            self._reset_tickets(max(splits, wsplits))
            self._set_chunk_keys(splits, max_len)
            if self.sliding_window_size:
                cpu = forward_batch.seq_lens_cpu.tolist() if forward_batch.seq_lens_cpu is not None else None
                md.window_kv_indptr, md.window_kv_indices = self._window(
                    forward_batch.seq_lens, cpu, forward_batch.req_pool_indices, bs)
            self.forward_metadata = md
            return
        # extend: prefix indices gathered once for all layers (triton_backend.py:283-320)
        dev = self.device
        kv_indptr = torch.zeros(bs + 1, dtype=torch.int32, device=dev)
        kv_indptr[1:] = torch.cumsum(forward_batch.extend_prefix_lens, dim=0)
        qo_indptr = torch.zeros(bs + 1, dtype=torch.int32, device=dev)
        qo_indptr[1:] = torch.cumsum(forward_batch.extend_seq_lens, dim=0)
        prefix_sum = sum(forward_batch.extend_prefix_lens_cpu)
        if prefix_sum > 0:
            kv_indices = torch.empty(prefix_sum, dtype=torch.int32, device=dev)
            ops.create_kv_indices(self.req_to_token, forward_batch.req_pool_indices,
                                  forward_batch.extend_prefix_lens, kv_indptr, None, kv_indices)
        else:
            kv_indices = self._dummy_idx
        md = ForwardMetadata(qo_indptr=qo_indptr, kv_indptr=kv_indptr, kv_indices=kv_indices,
                             max_extend_len=max(forward_batch.extend_seq_lens_cpu))
        if self.sliding_window_size:
            md.window_kv_indptr, md.window_kv_indices = self._window(
                forward_batch.extend_prefix_lens, forward_batch.extend_prefix_lens_cpu, forward_batch.req_pool_indices, bs)
        self.forward_metadata = md

    # ---- HIP-graph capture / replay (static shapes; nothing to refresh per replay) ----
    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int):
        self._scratch(max_bs, MAX_KV_SPLITS)
        if self.sliding_window_size:
            self.graph_window_indptr = torch.zeros(max_bs + 1, dtype=torch.int32, device=self.device)
            self.graph_window_indices = torch.zeros(max_num_tokens * (self.sliding_window_size + 1), dtype=torch.int32,
                                                    device=self.device)

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens,
                                                 encoder_lens, forward_mode, spec_info):
        assert forward_mode.is_decode_or_idle(), "only decode is captured"
        # one graph serves every step: split for the longest context the model can reach
        splits = self._plan_splits(bs, self.max_context_len)
        self.graph_num_splits[bs] = splits
        wsplits = self._plan_splits(bs, self.sliding_window_size + 1) if self.sliding_window_size else 1
        logits, lse = self._scratch(bs, max(splits, wsplits)) if max(splits, wsplits) > 1 else (None, None)
        md = ForwardMetadata(num_kv_splits=splits, attn_logits=logits, attn_lse=lse, window_num_kv_splits=wsplits,
                             tickets=self._tickets if logits is not None else None,
                             kv_chunk_keys=self._chunk_keys if splits > 1 else None)
        self._set_chunk_keys(splits, 1)   # the capture runs on fill-value lengths
        if self.sliding_window_size:
            md.window_kv_indptr, md.window_kv_indices = self._window(
                seq_lens[:bs], None, req_pool_indices[:bs], bs, self.graph_window_indptr, self.graph_window_indices)
        self.forward_metadata = md

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum,
                                                encoder_lens, forward_mode, spec_info, seq_lens_cpu):
        # the captured kernels read req_to_token / seq_lens from the static buffers directly; only the
        # window gather of sliding-window models is refreshed (into the buffers the graph captured)
        if self.sliding_window_size:
            self._window(seq_lens[:bs], None, req_pool_indices[:bs], bs, self.graph_window_indptr, self.graph_window_indices)
        splits = self.graph_num_splits.get(bs, 1)
        self._reset_tickets(max(splits, 2 if self.sliding_window_size else 1))
        if splits > 1:
            # the graph was planned for the longest context the model can reach; the batch in front of us sets
            # the chunk, so that its longest sequence uses every split and shorter ones proportionally fewer
            if seq_lens_cpu is not None:
                self._set_chunk_keys(splits, int(seq_lens_cpu.max()) if bs > 0 else 1)
            else:
                self._chunk_keys.copy_(((seq_lens[:bs].max() + (splits - 1)) // splits).clamp(min=1).to(torch.int32).view(1))

    def _set_chunk_keys(self, splits: int, max_len: int):
        if splits > 1:
            self._chunk_keys.fill_(max(1, (max_len + splits - 1) // splits))

    def _reset_tickets(self, splits: int):
        """The kernels hand the split-merge counters back zeroed; a launch that died half-way (a failed
        step, an aborted capture) would leave some raised and every later merge of that cell would never
        fire.  One small fill per forward makes the next step start clean whatever happened before."""
        if splits > 1 and self._tickets is not None:
            self._tickets.zero_()
            # graphs captured before the array was outgrown still count in the old one
            for t in self._retired_tickets:
                t.zero_()

    def get_cuda_graph_seq_len_fill_value(self):
        return 1

    # ---- per layer ----
    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        ops = get_ops()
        q = q.reshape(-1, layer.tp_q_head_num * layer.qk_head_dim)
        o = q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim))
        pool = forward_batch.token_to_kv_pool
        if save_kv_cache:
            pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v)
        md = self.forward_metadata
        if hasattr(pool, "kv_lora_rank"):
            # MLA: one latent row per token serves as key (rank + rope) and value (rank) of every
            # head (deepseek_v2.py:866-875 attn_mqa, num_kv_heads = 1)
            ops.decode_attention_mla(
                q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), pool.get_key_buffer(layer.layer_id),
                o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
                forward_batch.req_to_token_pool.req_to_token, forward_batch.req_pool_indices,
                forward_batch.seq_lens, md.attn_logits, md.attn_lse, md.num_kv_splits, pool.kv_lora_rank,
                layer.scaling, layer.logit_cap, tickets=md.tickets, kv_chunk_keys=md.kv_chunk_keys)
            return o
        if self._windowed(layer):
            ops.decode_attention_fwd(
                q.view(-1, layer.tp_q_head_num, layer.qk_head_dim),
                pool.get_key_buffer(layer.layer_id), pool.get_value_buffer(layer.layer_id),
                o.view(-1, layer.tp_q_head_num, layer.v_head_dim), md.window_kv_indptr, md.window_kv_indices,
                md.attn_logits, md.attn_lse, md.window_num_kv_splits, MAX_KV_SPLITS, layer.scaling, layer.logit_cap,
                tickets=md.tickets)
            return o
        ops.decode_attention_paged(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim),
            pool.get_key_buffer(layer.layer_id), pool.get_value_buffer(layer.layer_id),
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            forward_batch.req_to_token_pool.req_to_token, forward_batch.req_pool_indices,
            forward_batch.seq_lens, md.attn_logits, md.attn_lse, md.num_kv_splits,
            layer.scaling, layer.logit_cap, tickets=md.tickets, kv_chunk_keys=md.kv_chunk_keys)
        return o

    def _windowed(self, layer) -> bool:
        """triton_backend.py:656, 711: a layer with sliding_window_size > -1 attends the window lists."""
        sw = layer.sliding_window_size
        if sw is None or sw <= -1:
            return False
        if self.forward_metadata.window_kv_indptr is None or sw != self.sliding_window_size:
            raise NotImplementedError(
                f"layer {layer.layer_id} has sliding_window_size {sw} but the backend was built for "
                f"model_runner.sliding_window_size = {self.sliding_window_size}")
        return True

    def forward_decode_fused_qkv(self, qkv_part, positions, cos_sin_cache, layer, forward_batch,
                                 part_x_scale=None, part_w_scale=None, out_dtype=None):
        """Decode step from the qkv GEMM's split-K partials: reduction, RoPE, KV-cache write and
        attention in one launch (ops.decode_attention_fused_qkv).  Same result as rotary_emb +
        forward_decode(save_kv_cache=True) on the reduced projection."""
        o = torch.empty((qkv_part.shape[1], layer.tp_q_head_num * layer.v_head_dim),
                        dtype=out_dtype or forward_batch.token_to_kv_pool.dtype, device=qkv_part.device)
        pool = forward_batch.token_to_kv_pool
        md = self.forward_metadata
        get_ops().decode_attention_fused_qkv(
            qkv_part, positions, forward_batch.out_cache_loc, cos_sin_cache,
            pool.get_key_buffer(layer.layer_id), pool.get_value_buffer(layer.layer_id),
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            forward_batch.req_to_token_pool.req_to_token, forward_batch.req_pool_indices,
            forward_batch.seq_lens, md.attn_logits, md.attn_lse, md.num_kv_splits, layer.scaling,
            part_x_scale=part_x_scale, part_w_scale=part_w_scale, tickets=md.tickets, kv_chunk_keys=md.kv_chunk_keys)
        return o

    def supports_mla_fused_rope(self, layer, rotary_emb, forward_batch, dtype) -> bool:
        """Whether forward_decode_mla_fused_rope serves this decode step (MLA_FUSED_ROPE of the reference,
        deepseek_v2.py:973-980: ROCm, decode only)."""
        pool = forward_batch.token_to_kv_pool
        return (MLA_FUSED_ROPE and forward_batch.forward_mode.is_decode() and hasattr(pool, "kv_lora_rank")
                and rotary_emb.rotary_dim == pool.qk_rope_head_dim
                and rotary_emb.cos_sin_cache.dtype in (torch.float32, dtype) and pool.dtype in (dtype, torch.float8_e4m3fn)
                and get_ops().decode_attention_mla_fused_rope_supported(pool.kv_lora_rank, pool.qk_rope_head_dim, dtype,
                                                                         pool.dtype))

    def forward_decode_mla_fused_rope(self, q, k_nope, k_pe, positions, rotary_emb, layer, forward_batch):
        """Absorbed MLA decode from the UNROTATED q rope part / k_pe: RoPE, the write of the new latent row
        and the attention in one launch.  Same result as rotary_emb + set_kv_buffer + forward_decode."""
        pool = forward_batch.token_to_kv_pool
        md = self.forward_metadata
        q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
        o = q.new_empty((q3.shape[0], layer.tp_q_head_num * layer.v_head_dim))
        get_ops().decode_attention_mla_fused_rope(
            q3, k_nope, k_pe, positions.flatten(), forward_batch.out_cache_loc, rotary_emb.cos_sin_cache,
            rotary_emb.is_neox_style, pool.get_key_buffer(layer.layer_id), o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            forward_batch.req_to_token_pool.req_to_token, forward_batch.req_pool_indices, forward_batch.seq_lens,
            md.attn_logits, md.attn_lse, md.num_kv_splits, pool.kv_lora_rank, layer.scaling, layer.logit_cap,
            tickets=md.tickets, kv_chunk_keys=md.kv_chunk_keys)
        return o

    @staticmethod
    def supports_fused_qkv(layer, rotary_emb, dtype) -> bool:
        g = layer.tp_q_head_num // max(layer.tp_k_head_num, 1)
        return (layer.qk_head_dim == 128 and layer.v_head_dim == 128 and dtype in (torch.bfloat16, torch.float16)
                and layer.tp_q_head_num % layer.tp_k_head_num == 0 and g in (1, 2, 4, 8)
                and not layer.logit_cap and layer.sliding_window_size in (None, -1)
                and rotary_emb.is_neox_style and rotary_emb.rotary_dim == 128
                and rotary_emb.cos_sin_cache.dtype in (torch.float32, dtype))

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        ops = get_ops()
        o = q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim))
        pool = forward_batch.token_to_kv_pool
        if save_kv_cache:
            pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v)
        md = self.forward_metadata
        if hasattr(pool, "kv_lora_rank") and layer.qk_head_dim == pool.kv_lora_rank + pool.qk_rope_head_dim:
            # MLA in the absorbed form (attn_mqa) over a cached prefix: the new latent rows are in
            # the cache by now (written above), so token t of a request is a decode query over the
            # request's first position(t) + 1 rows -- the multi-query kernel of the decode path
            # serves it with per-token table rows and lengths.  Causal by construction.
            assert save_kv_cache, "the absorbed MLA extend reads the new tokens from the cache"
            q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
            if (MLA_EXTEND_TILED and pool.dtype != torch.float8_e4m3fn   # (the tiled kernel lands raw 16-bit rows)
                    and (q3.shape[0] // 4) * ((layer.tp_q_head_num + 15) // 16) >= MLA_EXTEND_TILED_MIN_WGS
                    and ops.extend_attention_mla_supported(layer.tp_q_head_num, pool.kv_lora_rank,
                                                           pool.qk_rope_head_dim, q.dtype)):
                # four consecutive tokens of a request share one pass over the rows
                ops.extend_attention_mla(
                    q3, pool.get_key_buffer(layer.layer_id), o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
                    forward_batch.req_to_token_pool.req_to_token, forward_batch.req_pool_indices,
                    forward_batch.seq_lens, md.qo_indptr, md.max_extend_len, pool.kv_lora_rank, layer.scaling,
                    layer.logit_cap)
                return o
            tok_req = torch.repeat_interleave(forward_batch.req_pool_indices, forward_batch.extend_seq_lens.to(torch.int64))
            tok_len = (forward_batch.positions + 1).to(torch.int64)
            ops.decode_attention_mla(
                q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), pool.get_key_buffer(layer.layer_id),
                o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
                forward_batch.req_to_token_pool.req_to_token, tok_req, tok_len, None, None, 1, pool.kv_lora_rank,
                layer.scaling, layer.logit_cap)
            return o
        if hasattr(pool, "kv_lora_rank") and sum(forward_batch.extend_prefix_lens_cpu) > 0:
            raise NotImplementedError("MLA multi-head prefill cannot read a cached prefix: use the absorbed form (attn_mqa)")
        windowed = self._windowed(layer)
        if hasattr(pool, "kv_lora_rank"):
            # multi-head prefill over an MLA pool: there is no prefix (checked above), so the cache buffers are
            # never read -- and they are latent rows (possibly e4m3fn bytes), not per-head keys.  The kernel
            # still wants valid pointers: hand it the new tokens' own keys / values.
            kbuf = k.reshape(k.shape[0], -1, layer.qk_head_dim).contiguous()
            vbuf = v.reshape(v.shape[0], -1, layer.v_head_dim).contiguous()
        else:
            kbuf, vbuf = pool.get_key_buffer(layer.layer_id), pool.get_value_buffer(layer.layer_id)
        ops.extend_attention_fwd(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), k, v,
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            kbuf, vbuf,
            md.qo_indptr, md.window_kv_indptr if windowed else md.kv_indptr,
            md.window_kv_indices if windowed else md.kv_indices, None, True, None, md.max_extend_len,
            layer.scaling, layer.logit_cap, sliding_window_size=layer.sliding_window_size if windowed else -1)
        return o
