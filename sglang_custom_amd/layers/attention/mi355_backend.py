"""Mi355AttnBackend: hand-written HIP attention behind the AttentionBackend surface.

Plays the role of TritonAttnBackend (python/sglang/srt/layers/attention/triton_backend.py:
init_forward_metadata :160-336, graph variants :338-627, forward_extend :632-685,
forward_decode :687-732) with two deliberate differences:
  * decode reads the req_to_token page table directly (as the reference's CPU kernel does,
    sgl-kernel/csrc/cpu/decode.cpp:913-916), so no kv_indices gather and no per-forward
    allocation or device->host read is needed: metadata is a split count chosen on the host;
  * the per-forward scratch (split partials) lives in grow-only buffers owned by the backend.
Called once per layer; the backend owns the KV write when save_kv_cache is set.
"""
import dataclasses
from typing import Optional

import torch

from ...ops_backend import get_ops
from .base_attn_backend import AttentionBackend

MAX_KV_SPLITS = 64


@dataclasses.dataclass
class ForwardMetadata:
    num_kv_splits: int = 1
    attn_logits: Optional[torch.Tensor] = None
    attn_lse: Optional[torch.Tensor] = None
    qo_indptr: Optional[torch.Tensor] = None
    kv_indptr: Optional[torch.Tensor] = None
    kv_indices: Optional[torch.Tensor] = None
    max_extend_len: int = 0


class Mi355AttnBackend(AttentionBackend):
    def __init__(self, model_runner):
        super().__init__()
        self.device = model_runner.device
        self.req_to_token = model_runner.req_to_token_pool.req_to_token
        cfg = model_runner.model_config
        tp = model_runner.tp_size
        self.num_head = cfg.num_attention_heads // tp
        self.num_kv_head = max(1, cfg.num_key_value_heads // tp)
        self.v_head_dim = model_runner.token_to_kv_pool.v_head_dim
        self.forward_metadata: Optional[ForwardMetadata] = None
        self._logits = None
        self._lse = None
        self._dummy_idx = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.graph_num_splits = {}

    # ---- scratch ----
    def _scratch(self, bs: int, splits: int):
        need = bs * self.num_head * splits
        if self._lse is None or self._lse.numel() < need:
            self._logits = torch.empty(need * self.v_head_dim, dtype=torch.float32, device=self.device)
            self._lse = torch.empty(need, dtype=torch.float32, device=self.device)
        return self._logits, self._lse

    def _plan_splits(self, bs: int, max_seq_len: int) -> int:
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
            logits, lse = self._scratch(bs, splits) if splits > 1 else (None, None)
            self.forward_metadata = ForwardMetadata(num_kv_splits=splits, attn_logits=logits, attn_lse=lse)
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
        self.forward_metadata = ForwardMetadata(
            qo_indptr=qo_indptr, kv_indptr=kv_indptr, kv_indices=kv_indices,
            max_extend_len=max(forward_batch.extend_seq_lens_cpu))

    # ---- HIP-graph capture / replay (static shapes; nothing to refresh per replay) ----
    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int):
        self._scratch(max_bs, MAX_KV_SPLITS)

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens,
                                                 encoder_lens, forward_mode, spec_info,
                                                 max_seq_len: int = 4096):
        assert forward_mode.is_decode_or_idle(), "only decode is captured"
        splits = self._plan_splits(bs, max_seq_len)
        self.graph_num_splits[bs] = splits
        logits, lse = self._scratch(bs, splits) if splits > 1 else (None, None)
        self.forward_metadata = ForwardMetadata(num_kv_splits=splits, attn_logits=logits, attn_lse=lse)

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum,
                                                encoder_lens, forward_mode, spec_info, seq_lens_cpu):
        # the captured kernels read req_to_token / seq_lens from the static buffers directly
        return

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
                layer.scaling, layer.logit_cap)
            return o
        ops.decode_attention_paged(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim),
            pool.get_key_buffer(layer.layer_id), pool.get_value_buffer(layer.layer_id),
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            forward_batch.req_to_token_pool.req_to_token, forward_batch.req_pool_indices,
            forward_batch.seq_lens, md.attn_logits, md.attn_lse, md.num_kv_splits,
            layer.scaling, layer.logit_cap)
        return o

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
            part_x_scale=part_x_scale, part_w_scale=part_w_scale)
        return o

    @staticmethod
    def supports_fused_qkv(layer, rotary_emb, dtype) -> bool:
        g = layer.tp_q_head_num // max(layer.tp_k_head_num, 1)
        return (hasattr(get_ops(), "decode_attention_fused_qkv") and layer.qk_head_dim == 128
                and layer.v_head_dim == 128 and dtype in (torch.bfloat16, torch.float16)
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
        if hasattr(pool, "kv_lora_rank") and layer.qk_head_dim == pool.head_dim:
            # MLA in the absorbed form (attn_mqa) over a cached prefix: the new latent rows are in
            # the cache by now (written above), so token t of a request is a decode query over the
            # request's first position(t) + 1 rows -- the multi-query kernel of the decode path
            # serves it with per-token table rows and lengths.  Causal by construction.
            assert save_kv_cache, "the absorbed MLA extend reads the new tokens from the cache"
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
        ops.extend_attention_fwd(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), k, v,
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            pool.get_key_buffer(layer.layer_id),
            # (an MLA pool has no separate value rows; without a prefix neither buffer is read)
            pool.get_key_buffer(layer.layer_id) if hasattr(pool, "kv_lora_rank") else pool.get_value_buffer(layer.layer_id),
            md.qo_indptr, md.kv_indptr, md.kv_indices, None, True, None, md.max_extend_len,
            layer.scaling, layer.logit_cap, sliding_window_size=layer.sliding_window_size)
        return o
