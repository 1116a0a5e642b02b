"""AttentionBackend: the plug-in surface of the reference
(python/sglang/srt/layers/attention/base_attn_backend.py:14-115), method for method."""
from abc import ABC, abstractmethod
from typing import Optional

import torch


class AttentionBackend(ABC):
    @abstractmethod
    def init_forward_metadata(self, forward_batch):
        """Init the metadata for a forward pass (called once per forward, not per layer)."""
        raise NotImplementedError()

    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int):
        raise NotImplementedError()

    def init_forward_metadata_capture_cuda_graph(self, bs: int, num_tokens: int,
                                                 req_pool_indices: torch.Tensor,
                                                 seq_lens: torch.Tensor,
                                                 encoder_lens: Optional[torch.Tensor],
                                                 forward_mode, spec_info):
        raise NotImplementedError()

    def init_forward_metadata_replay_cuda_graph(self, bs: int, req_pool_indices: torch.Tensor,
                                                seq_lens: torch.Tensor, seq_lens_sum: int,
                                                encoder_lens: Optional[torch.Tensor], forward_mode,
                                                spec_info, seq_lens_cpu: Optional[torch.Tensor]):
        raise NotImplementedError()

    def get_cuda_graph_seq_len_fill_value(self):
        raise NotImplementedError()

    def forward(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True, **kwargs):
        if forward_batch.forward_mode.is_decode():
            return self.forward_decode(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache,
                                       **kwargs)
        return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache,
                                   **kwargs)

    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        raise NotImplementedError()

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        raise NotImplementedError()

    def support_triton(self):
        return False

    # ---- extension beyond the reference's surface (optional; the default declines) ----
    @staticmethod
    def supports_fused_qkv(layer, rotary_emb, dtype) -> bool:
        """Whether forward_decode_fused_qkv (decode attention straight from the qkv projection's
        split-K partial sums: reduction, RoPE, KV-cache write and attention in one launch) serves this
        layer.  Backends that do not have it inherit this False and the model takes the plain
        qkv -> rotary_emb -> forward_decode sequence."""
        return False

    def forward_decode_fused_qkv(self, qkv_part, positions, cos_sin_cache, layer, forward_batch,
                                 part_x_scale=None, part_w_scale=None, out_dtype=None):
        raise NotImplementedError()

    def supports_mla_fused_rope(self, layer, rotary_emb, forward_batch, dtype) -> bool:
        """Whether forward_decode_mla_fused_rope (MLA decode that rotates q / k_pe and writes the new latent
        row inside the attention launch) serves this step; the default declines."""
        return False

    def forward_decode_mla_fused_rope(self, q, k_nope, k_pe, positions, rotary_emb, layer, forward_batch):
        raise NotImplementedError()

