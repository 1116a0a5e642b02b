"""Multi-head latent attention of DeepSeek-V2/V3: the attention module only (tensor-parallel size 1),
with the reference's two forward methods (python/sglang/srt/models/deepseek_v2.py):

  forward_normal  (:1131-1179)  prefill without a cached prefix: the latent is expanded through
                  kv_b_proj into per-head keys (nope 128 + rope 64) and values (128) and runs as
                  ordinary multi-head attention; what is WRITTEN to the cache is the latent row
                  [normed c_kv (kv_lora_rank) | roped k_pe (qk_rope_head_dim)], one per token.
  forward_absorb  (:1181-1335)  decode: kv_b_proj is folded into the query (q_nope . w_kc) and the
                  output (. w_vc), so attention is multi-query over the latent cache: 576-wide keys,
                  512-wide values, one KV "head" shared by all heads (mi355_decode_attention_mla).

dispatch_attn_forward_method (:969-1039) picks between them as the reference's Triton backend does:
extend without prefix -> normal; decode and extend over a cached prefix -> absorb (the latter as one
decode-style multi-query pass per new token over the rows cached so far: exact, O(new x cached) reads).
With rope_scaling (DeepSeek's YaRN) the rotary table is the interpolated, magnitude-scaled one and the
softmax scale is multiplied by mscale(factor, mscale_all_dim)^2 (:845-862); the rotation is the
interleaved (GPT-J) layout.
"""
from typing import Optional

import torch
from torch import nn

from ..layers.layernorm import RMSNorm
from ..layers.linear import ReplicatedLinear
from ..layers.radix_attention import RadixAttention
from ..layers.rotary_embedding import get_rope, yarn_get_mscale


class DeepseekV2AttentionMLA(nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, qk_nope_head_dim: int, qk_rope_head_dim: int,
                 v_head_dim: int, q_lora_rank: Optional[int], kv_lora_rank: int, layer_id: int, dtype, device,
                 rope_theta: float = 10000.0, max_position_embeddings: int = 8192, rms_norm_eps: float = 1e-6,
                 rope_scaling: Optional[dict] = None):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_local_heads = num_heads
        self.qk_nope_head_dim, self.qk_rope_head_dim = qk_nope_head_dim, qk_rope_head_dim
        self.qk_head_dim = qk_nope_head_dim + qk_rope_head_dim
        self.v_head_dim = v_head_dim
        self.q_lora_rank, self.kv_lora_rank = q_lora_rank, kv_lora_rank
        self.scaling = self.qk_head_dim ** -0.5
        if q_lora_rank is not None:
            # fused_qkv_a_proj_with_mqa: [q_a | kv_a + k_pe] in one projection (:760-768)
            self.fused_qkv_a_proj_with_mqa = ReplicatedLinear(
                hidden_size, q_lora_rank + kv_lora_rank + qk_rope_head_dim, dtype, device)
            self.q_a_layernorm = RMSNorm(q_lora_rank, eps=rms_norm_eps)
            self.q_b_proj = ReplicatedLinear(q_lora_rank, num_heads * self.qk_head_dim, dtype, device)
        else:
            self.q_proj = ReplicatedLinear(hidden_size, num_heads * self.qk_head_dim, dtype, device)
            self.kv_a_proj_with_mqa = ReplicatedLinear(hidden_size, kv_lora_rank + qk_rope_head_dim, dtype, device)
        self.kv_a_layernorm = RMSNorm(kv_lora_rank, eps=rms_norm_eps)
        self.kv_b_proj = ReplicatedLinear(kv_lora_rank, num_heads * (qk_nope_head_dim + v_head_dim), dtype, device)
        self.o_proj = ReplicatedLinear(num_heads * v_head_dim, hidden_size, dtype, device)
        if rope_scaling:
            rope_scaling = dict(rope_scaling, rope_type="deepseek_yarn")
            mscale = yarn_get_mscale(rope_scaling["factor"], float(rope_scaling.get("mscale_all_dim", False)))
            self.scaling = self.scaling * mscale * mscale
        self.rotary_emb = get_rope(qk_rope_head_dim, qk_rope_head_dim, max_position_embeddings, rope_theta,
                                   is_neox_style=False, rope_scaling=rope_scaling, dtype=dtype)
        # the two RadixAttention views of one layer (:866-886)
        self.attn_mqa = RadixAttention(num_heads, kv_lora_rank + qk_rope_head_dim, self.scaling, num_kv_heads=1,
                                       layer_id=layer_id, v_head_dim=kv_lora_rank)
        self.attn_mha = RadixAttention(num_heads, self.qk_head_dim, self.scaling, num_kv_heads=num_heads,
                                       layer_id=layer_id, v_head_dim=v_head_dim)
        self.w_kc = self.w_vc = None
        for m in self.modules():
            if isinstance(m, RMSNorm):
                m.weight.data = m.weight.data.to(dtype=dtype, device=device)
        self.to(device)

    def load_kv_b_proj_fp8(self, weight: torch.Tensor, weight_scale: torch.Tensor) -> None:
        """kv_b_proj as an fp8 W8A8 checkpoint stores it: e4m3fn [N, K] + per-output-channel scale [N, 1].
        The absorbed (decode) path is built from it by post_load_weights(); the multi-head prefill, which
        would run kv_b_proj as an fp8 linear, is outside this module's fp8 support."""
        assert weight.dtype == torch.float8_e4m3fn and weight_scale.shape == (weight.shape[0], 1)
        self._kv_b_fp8 = (weight, weight_scale.float())

    def post_load_weights(self) -> None:
        """w_kc [H, nope, rank] and w_vc [H, rank, v]: kv_b_proj split per head (:2150-2313).  From an
        fp8 kv_b_proj the reference re-quantises to ONE scale per tensor (channel_quant_to_tensor_quant,
        :2251) and, on ROCm, multiplies as w.to(bf16) * w_scale on every call (:1242-1247, 1300-1306);
        the same two bf16 tensors are built once here."""
        fp8 = getattr(self, "_kv_b_fp8", None)
        w = self.kv_b_proj.weight.data
        if fp8 is not None:
            wq, ws = fp8
            x = wq.to(torch.float32) * ws
            lo, hi = x.aminmax()
            amax = torch.maximum(lo.abs(), hi.abs()).float().clamp(min=1e-12)
            fp_max = torch.finfo(torch.float8_e4m3fn).max
            scale = fp_max / amax
            w = (x.float() * scale).clamp(min=-fp_max, max=fp_max).to(torch.float8_e4m3fn)
            self.w_scale = scale.float().reciprocal()
        w_kc, w_vc = w.unflatten(0, (-1, self.qk_nope_head_dim + self.v_head_dim)).split(
            [self.qk_nope_head_dim, self.v_head_dim], dim=1)
        if fp8 is not None:
            w_kc = w_kc.to(torch.bfloat16) * self.w_scale
            w_vc = w_vc.to(torch.bfloat16) * self.w_scale
        self.w_kc = w_kc.contiguous().to(self.kv_b_proj.weight.dtype)
        self.w_vc = w_vc.transpose(1, 2).contiguous().to(self.kv_b_proj.weight.dtype)

    # ---- shared front end ----
    def _q_and_latent(self, hidden_states):
        if self.q_lora_rank is not None:
            q, latent = self.fused_qkv_a_proj_with_mqa(hidden_states)[0].split(
                [self.q_lora_rank, self.kv_lora_rank + self.qk_rope_head_dim], dim=-1)
            q = self.q_b_proj(self.q_a_layernorm(q.contiguous()))[0]
        else:
            q = self.q_proj(hidden_states)[0]
            latent = self.kv_a_proj_with_mqa(hidden_states)[0]
        return q.view(-1, self.num_local_heads, self.qk_head_dim), latent

    def _rope(self, positions, q, latent):
        """Rotate the rope parts: q[..., nope:] in place; returns k_pe [T, 1, rope] rotated."""
        q_pe = q[..., self.qk_nope_head_dim:].contiguous()
        k_pe = latent[..., self.kv_lora_rank:].contiguous().unsqueeze(1)
        T = q.shape[0]
        q_pe, k_pe = self.rotary_emb(positions, q_pe.view(T, -1), k_pe.view(T, -1))
        q[..., self.qk_nope_head_dim:] = q_pe.view(T, self.num_local_heads, self.qk_rope_head_dim)
        return k_pe.view(T, 1, self.qk_rope_head_dim)

    # ---- prefill without prefix ----
    def forward_normal(self, positions, hidden_states, forward_batch):
        q, latent = self._q_and_latent(hidden_states)
        kv_a = self.kv_a_layernorm(latent[..., : self.kv_lora_rank].contiguous())
        kv = self.kv_b_proj(kv_a)[0].view(-1, self.num_local_heads, self.qk_nope_head_dim + self.v_head_dim)
        k_pe = self._rope(positions, q, latent)
        k = torch.empty_like(q)
        k[..., : self.qk_nope_head_dim] = kv[..., : self.qk_nope_head_dim]
        k[..., self.qk_nope_head_dim:] = k_pe
        v = kv[..., self.qk_nope_head_dim:].contiguous()
        latent_cache = torch.cat([kv_a.unsqueeze(1), k_pe], dim=-1)
        forward_batch.token_to_kv_pool.set_kv_buffer(self.attn_mha, forward_batch.out_cache_loc, latent_cache, None)
        attn_output = self.attn_mha(q, k, v, forward_batch, save_kv_cache=False)
        return self.o_proj(attn_output.reshape(-1, self.num_local_heads * self.v_head_dim))[0]

    # ---- decode ----
    def forward_absorb(self, positions, hidden_states, forward_batch):
        assert self.w_kc is not None, "call post_load_weights() after loading kv_b_proj"
        q, latent = self._q_and_latent(hidden_states)
        k_nope = self.kv_a_layernorm(latent[..., : self.kv_lora_rank].contiguous()).unsqueeze(1)
        k_pe = self._rope(positions, q, latent)
        q_nope, q_pe = q.split([self.qk_nope_head_dim, self.qk_rope_head_dim], dim=-1)
        q_nope_out = torch.bmm(q_nope.transpose(0, 1), self.w_kc).transpose(0, 1)       # [T, H, rank]
        qa = torch.cat([q_nope_out, q_pe], dim=-1)
        ka = torch.cat([k_nope, k_pe], dim=-1)
        attn_output = self.attn_mqa(qa, ka, k_nope, forward_batch)
        attn_output = attn_output.view(-1, self.num_local_heads, self.kv_lora_rank)
        out = torch.bmm(attn_output.transpose(0, 1), self.w_vc).transpose(0, 1).flatten(1, 2)
        return self.o_proj(out.contiguous())[0]

    def forward(self, positions, hidden_states, forward_batch):
        # dispatch_attn_forward_method (:969-1039) for the Triton backend: multi-head form for a
        # prefill without prefix, absorbed form otherwise (decode, or extend over cached latent rows)
        if forward_batch.forward_mode.is_decode() or sum(forward_batch.extend_prefix_lens_cpu) > 0:
            return self.forward_absorb(positions, hidden_states, forward_batch)
        return self.forward_normal(positions, hidden_states, forward_batch)
