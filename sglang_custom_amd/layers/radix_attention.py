"""RadixAttention: the per-layer hand-off to the attention backend
(python/sglang/srt/layers/radix_attention.py:59-108).  Holds the head geometry the backend
reads (tp_q_head_num, tp_k_head_num, head_dim, v_head_dim, scaling, layer_id, logit_cap,
sliding_window_size) and views k, v as [T, Hkv, D] before calling
forward_batch.attn_backend.forward."""
from torch import nn


class RadixAttention(nn.Module):
    def __init__(self, num_heads: int, head_dim: int, scaling: float, num_kv_heads: int,
                 layer_id: int, logit_cap: float = 0.0, v_head_dim: int = -1,
                 sliding_window_size: int = -1):
        super().__init__()
        self.tp_q_head_num = num_heads
        self.tp_k_head_num = num_kv_heads
        self.tp_v_head_num = num_kv_heads
        self.head_dim = head_dim
        self.qk_head_dim = head_dim
        self.v_head_dim = v_head_dim if v_head_dim != -1 else head_dim
        self.scaling = scaling
        self.layer_id = layer_id
        self.logit_cap = logit_cap
        self.sliding_window_size = sliding_window_size or -1
        self.k_scale = None
        self.v_scale = None

    def forward(self, q, k, v, forward_batch, save_kv_cache: bool = True, **kwargs):
        if k is not None:
            assert v is not None
            k = k.view(-1, self.tp_k_head_num, self.qk_head_dim)
            v = v.view(-1, self.tp_v_head_num, self.v_head_dim)
        return forward_batch.attn_backend.forward(q, k, v, self, forward_batch, save_kv_cache,
                                                  **kwargs)
