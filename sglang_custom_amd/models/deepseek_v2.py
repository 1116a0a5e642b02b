"""DeepSeek-V2/V3 with multi-head latent attention (tensor-parallel size 1): the attention module with
the reference's two forward methods (python/sglang/srt/models/deepseek_v2.py), and -- at the end of
the file -- the decoder stack around it for configurations whose layers are all dense
(first_k_dense_replace >= num_hidden_layers or no routed experts; :1826-1832 picks DeepseekV2MLP for
those).  Mixture-of-experts layers are outside this path and are refused at construction.

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
from typing import Dict, Optional

import torch
from torch import nn

from ..layers.activation import SiluAndMul
from ..layers.layernorm import RMSNorm
from ..distributed.parallel_state import get_tensor_model_parallel_world_size
from ..layers.linear import MergedColumnParallelLinear, ReplicatedLinear, RowParallelLinear
from ..layers.logits_processor import LogitsProcessor
from ..layers.quantization.unquant import (
    QuantizedAct,
    UnquantizedLinearMethod,
    decode_linear_deferred,
    pack_gate_up_for_decode,
)
from ..layers.radix_attention import RadixAttention
from ..ops_backend import get_ops
from ..layers.rotary_embedding import get_rope, yarn_get_mscale
from ..layers.vocab_parallel_embedding import ParallelLMHead, VocabParallelEmbedding
from .llama import LlamaMLP, takes_fp8_rows_of_k


class DeepseekV2AttentionMLA(nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, qk_nope_head_dim: int, qk_rope_head_dim: int,
                 v_head_dim: int, q_lora_rank: Optional[int], kv_lora_rank: int, layer_id: int, dtype, device,
                 rope_theta: float = 10000.0, max_position_embeddings: int = 8192, rms_norm_eps: float = 1e-6,
                 rope_scaling: Optional[dict] = None, quant_method_factory=None):
        super().__init__()
        qm = quant_method_factory or (lambda: None)   # one linear method per projection (fp8 W8A8 or unquantised)
        self.dtype = dtype
        self.hidden_size = hidden_size
        # Tensor parallelism as the reference shards this module (:781-840): the heads are divided over the
        # ranks -- q_b_proj / q_proj and kv_b_proj column-parallel, o_proj row-parallel with a sum over ranks --
        # while the low-rank "a" projections and the latent cache are replicated (one latent row per token
        # serves every head).
        tp = get_tensor_model_parallel_world_size()
        assert num_heads % tp == 0, f"{num_heads} heads over {tp} ranks"
        self.num_heads = num_heads
        self.num_local_heads = num_heads // tp
        self.qk_nope_head_dim, self.qk_rope_head_dim = qk_nope_head_dim, qk_rope_head_dim
        self.qk_head_dim = qk_nope_head_dim + qk_rope_head_dim
        self.v_head_dim = v_head_dim
        self.q_lora_rank, self.kv_lora_rank = q_lora_rank, kv_lora_rank
        self.scaling = self.qk_head_dim ** -0.5
        if q_lora_rank is not None:
            # fused_qkv_a_proj_with_mqa: [q_a | kv_a + k_pe] in one projection (:760-768)
            self.fused_qkv_a_proj_with_mqa = ReplicatedLinear(
                hidden_size, q_lora_rank + kv_lora_rank + qk_rope_head_dim, dtype, device, qm())
            self.q_a_layernorm = RMSNorm(q_lora_rank, eps=rms_norm_eps)
            self.q_b_proj = MergedColumnParallelLinear(q_lora_rank, [num_heads * self.qk_head_dim], dtype, device, qm())
        else:
            self.q_proj = MergedColumnParallelLinear(hidden_size, [num_heads * self.qk_head_dim], dtype, device, qm())
            self.kv_a_proj_with_mqa = ReplicatedLinear(hidden_size, kv_lora_rank + qk_rope_head_dim, dtype, device, qm())
        self.kv_a_layernorm = RMSNorm(kv_lora_rank, eps=rms_norm_eps)
        self.kv_b_proj = MergedColumnParallelLinear(kv_lora_rank, [num_heads * (qk_nope_head_dim + v_head_dim)], dtype, device, qm())
        self.o_proj = RowParallelLinear(num_heads * v_head_dim, hidden_size, dtype, device, qm())
        if rope_scaling:
            rope_scaling = dict(rope_scaling, rope_type="deepseek_yarn")
            mscale = yarn_get_mscale(rope_scaling["factor"], float(rope_scaling.get("mscale_all_dim", False)))
            self.scaling = self.scaling * mscale * mscale
        self.rotary_emb = get_rope(qk_rope_head_dim, qk_rope_head_dim, max_position_embeddings, rope_theta,
                                   is_neox_style=False, rope_scaling=rope_scaling, dtype=dtype)
        # the two RadixAttention views of one layer (:866-886)
        self.attn_mqa = RadixAttention(self.num_local_heads, kv_lora_rank + qk_rope_head_dim, self.scaling, num_kv_heads=1,
                                       layer_id=layer_id, v_head_dim=kv_lora_rank)
        self.attn_mha = RadixAttention(self.num_local_heads, self.qk_head_dim, self.scaling, num_kv_heads=self.num_local_heads,
                                       layer_id=layer_id, v_head_dim=v_head_dim)
        self.w_kc = self.w_vc = None
        self.w_kc_t = self.w_vc_t = None    # the same per-head matrices transposed, for the one-launch decode products
        for m in self.modules():
            if isinstance(m, RMSNorm):
                m.weight.data = m.weight.data.to(dtype=dtype, device=device)
        self.to(device)

    def load_kv_b_proj_fp8(self, weight: torch.Tensor, weight_scale: torch.Tensor) -> None:
        """kv_b_proj as an fp8 W8A8 checkpoint stores it: e4m3fn [N, K] + per-output-channel scale [N, 1].
        The absorbed (decode) path is built from it by post_load_weights(); the multi-head prefill, which
        would run kv_b_proj as an fp8 linear, is outside this module's fp8 support."""
        assert weight.dtype == torch.float8_e4m3fn and weight_scale.shape == (weight.shape[0], 1)
        if weight.shape[0] != self.kv_b_proj.weight.shape[0]:   # the whole checkpoint tensor: keep this rank's heads
            per, r = self.kv_b_proj.weight.shape[0], self.kv_b_proj.tp_rank
            assert weight.shape[0] == per * self.kv_b_proj.tp_size
            weight, weight_scale = weight[r * per:(r + 1) * per], weight_scale[r * per:(r + 1) * per]
        self._kv_b_fp8 = (weight, weight_scale.float())

    def linears(self):
        if self.q_lora_rank is not None:
            return [self.fused_qkv_a_proj_with_mqa, self.q_b_proj, self.kv_b_proj, self.o_proj]
        return [self.q_proj, self.kv_a_proj_with_mqa, self.kv_b_proj, self.o_proj]

    def post_load_weights(self) -> None:
        """w_kc [H, nope, rank] and w_vc [H, rank, v]: kv_b_proj split per head (:2150-2313).  From an
        fp8 kv_b_proj the reference re-quantises to ONE scale per tensor (channel_quant_to_tensor_quant,
        :2251) and, on ROCm, multiplies as w.to(bf16) * w_scale on every call (:1242-1247, 1300-1306);
        the same two bf16 tensors are built once here."""
        fp8 = getattr(self, "_kv_b_fp8", None)
        w = self.kv_b_proj.weight.data
        if fp8 is None and w.dtype == torch.float8_e4m3fn:   # quantised at load by the fp8 W8A8 linear method
            fp8 = (w, self.kv_b_proj.weight_scale.data.float())
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
        self.w_kc = w_kc.contiguous().to(self.dtype)
        self.w_vc = w_vc.transpose(1, 2).contiguous().to(self.dtype)
        if self.w_kc.is_cuda and self.dtype in (torch.bfloat16, torch.float16):
            # [H, n, k] with the reduction dimension contiguous: what ops.mla_absorb streams (2 x 2 MB per layer at
            # DeepSeek-V3's TP=8 shard shape)
            self.w_kc_t = self.w_kc.transpose(1, 2).contiguous()
            self.w_vc_t = self.w_vc.transpose(1, 2).contiguous()

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
        front = self._decode_front(hidden_states, max_rows=None)
        if front is not None:       # both latent norms and the k_pe copy in one launch, as in decode
            q, kv_a, k_pe_raw = front
            latent = None
        else:
            q, latent = self._q_and_latent(hidden_states)
            kv_a = self.kv_a_layernorm(latent[..., : self.kv_lora_rank].contiguous())
            k_pe_raw = latent[..., self.kv_lora_rank:]
        kv = self.kv_b_proj(kv_a)[0].view(-1, self.num_local_heads, self.qk_nope_head_dim + self.v_head_dim)
        ops, pool, rot = get_ops(), forward_batch.token_to_kv_pool, self.rotary_emb
        if (q.is_cuda and hasattr(pool, "kv_lora_rank") and rot.rotary_dim == self.qk_rope_head_dim
                and rot.cos_sin_cache.dtype in (torch.float32, q.dtype) and pool.dtype in (q.dtype, torch.float8_e4m3fn)
                and positions.dtype == torch.int64 and kv.is_contiguous() and q.is_contiguous()
                and ops.mla_prefill_assemble_supported(self.qk_nope_head_dim, self.qk_rope_head_dim, self.v_head_dim,
                                                       self.kv_lora_rank, q.dtype)):
            # RoPE of q and k_pe, the k / v assembly and the write of the latent rows in one launch
            k, v = ops.mla_prefill_assemble(
                q, kv, k_pe_raw, kv_a, positions.flatten(), rot.cos_sin_cache, rot.is_neox_style,
                pool.get_key_buffer(self.attn_mha.layer_id), forward_batch.out_cache_loc, self.qk_nope_head_dim,
                self.qk_rope_head_dim, self.v_head_dim)
            attn_output = self.attn_mha(q, k, v, forward_batch, save_kv_cache=False)
            return self.o_proj(attn_output.reshape(-1, self.num_local_heads * self.v_head_dim))[0]
        if latent is None:
            latent = torch.cat([kv_a, k_pe_raw], dim=-1)
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
    def _decode_front(self, hidden_states, max_rows=64):
        """Decode-sized batches (and, with max_rows=None, prefill batches from the projection's dense output): the "a" projection's split-K partials (or its output) go through ONE launch that
        sums them and applies q_a_layernorm and kv_a_layernorm (ops.mla_latent_norms) instead of a split, two
        copies and two norms.  Returns (q [T, H, nope + rope], normed c_kv [T, rank], k_pe [T, rope] unrotated),
        or None when the shape or the ops backend does not serve it."""
        ops = get_ops()
        x = hidden_states.dense if isinstance(hidden_states, QuantizedAct) else hidden_states   # (the norm's 16-bit rows)
        rank, rope = self.kv_lora_rank, self.qk_rope_head_dim
        q_rank = self.q_lora_rank or 0
        if (x.dim() != 2 or not x.is_cuda or x.shape[0] == 0 or (max_rows is not None and x.shape[0] > max_rows)
                or not ops.mla_latent_norms_supported(q_rank, rank, rope, x.dtype)):
            return None
        proj = self.fused_qkv_a_proj_with_mqa if q_rank else self.kv_a_proj_with_mqa
        qw = self.q_a_layernorm.weight.data if q_rank else None
        kvw, eps = self.kv_a_layernorm.weight.data, self.kv_a_layernorm.variance_epsilon
        q8 = bool(q_rank) and takes_fp8_rows_of_k(self.q_b_proj, x.shape[0])   # the launch also quantises q_a for an fp8 q_b_proj
        part = decode_linear_deferred(proj, hidden_states, needs_all_reduce=False)
        src = (dict(part=part.part, part_x_scale=part.x_scale, part_w_scale=part.w_scale) if part is not None
               else dict(x=proj(hidden_states)[0]))
        out = ops.mla_latent_norms(qw, kvw, q_rank, rank, rope, eps, quant_q_fp8=q8, **src)
        q, c_kv, k_pe = out[:3]
        if q8:
            q = QuantizedAct(out[3], out[4], q.dtype, q)
        q = self.q_b_proj(q)[0] if q_rank else self.q_proj(hidden_states)[0]
        return q.view(-1, self.num_local_heads, self.qk_head_dim), c_kv, k_pe

    def _absorb_q(self, q):
        """[T, H, nope + rope] -> [T, H, rank + rope]: q_nope . w_kc per head with q's rope part carried behind it
        (deepseek_v2.py:1234-1262)."""
        ops = get_ops()
        if (self.w_kc_t is not None and q.is_cuda
                and ops.mla_absorb_supported(q.shape[0], self.kv_lora_rank, self.qk_nope_head_dim, self.qk_rope_head_dim, q.dtype)):
            return ops.mla_absorb(q, self.w_kc_t, tail=self.qk_rope_head_dim)
        q_nope, q_pe = q.split([self.qk_nope_head_dim, self.qk_rope_head_dim], dim=-1)
        q_nope_out = torch.bmm(q_nope.transpose(0, 1), self.w_kc).transpose(0, 1)       # [T, H, rank]
        return torch.cat([q_nope_out, q_pe], dim=-1)

    def _absorb_out(self, attn_output):
        """[T, H * rank] -> [T, H * v]: attn . w_vc per head, in o_proj's input layout (deepseek_v2.py:1296-1333)."""
        ops = get_ops()
        a = attn_output.view(-1, self.num_local_heads, self.kv_lora_rank)
        if (self.w_vc_t is not None and a.is_cuda
                and ops.mla_absorb_supported(a.shape[0], self.v_head_dim, self.kv_lora_rank, 0, a.dtype)):
            return ops.mla_absorb(a, self.w_vc_t).flatten(1, 2)
        return torch.bmm(a.transpose(0, 1), self.w_vc).transpose(0, 1).flatten(1, 2).contiguous()

    def _o_proj(self, x):
        """o_proj with its split-K sum (tensor-parallel: its sum over the ranks) left to the norm that follows."""
        deferred = decode_linear_deferred(self.o_proj, x)
        if deferred is not None:
            return deferred
        return self.o_proj(x, defer_all_reduce=True)[0]

    def forward_absorb(self, positions, hidden_states, forward_batch):
        assert self.w_kc is not None, "call post_load_weights() after loading kv_b_proj"
        backend = forward_batch.attn_backend
        fused_rope = backend.supports_mla_fused_rope(self.attn_mqa, self.rotary_emb, forward_batch, hidden_states.dtype)
        front = self._decode_front(hidden_states) if forward_batch.forward_mode.is_decode() else None
        if front is not None:
            q, c_kv, k_pe = front
        else:
            q, latent = self._q_and_latent(hidden_states)
            c_kv = self.kv_a_layernorm(latent[..., : self.kv_lora_rank].contiguous())
            k_pe = latent[..., self.kv_lora_rank:]
        if fused_rope:
            # MLA_FUSED_ROPE (deepseek_v2.py:973-980, 1337-1536): q's rope part and k_pe go in unrotated; the
            # decode launch rotates them, writes the new latent row and attends
            attn_output = backend.forward_decode_mla_fused_rope(
                self._absorb_q(q), c_kv, k_pe, positions, self.rotary_emb, self.attn_mqa, forward_batch)
            return self._o_proj(self._absorb_out(attn_output))
        T = q.shape[0]
        q_pe = q[..., self.qk_nope_head_dim:].contiguous()
        q_pe, k_pe = self.rotary_emb(positions, q_pe.view(T, -1), k_pe.contiguous().view(T, -1))
        q[..., self.qk_nope_head_dim:] = q_pe.view(T, self.num_local_heads, self.qk_rope_head_dim)
        k_nope = c_kv.unsqueeze(1)
        ka = torch.cat([k_nope, k_pe.view(T, 1, self.qk_rope_head_dim)], dim=-1)
        attn_output = self.attn_mqa(self._absorb_q(q), ka, k_nope, forward_batch)
        return self._o_proj(self._absorb_out(attn_output))

    def forward(self, positions, hidden_states, forward_batch):
        # dispatch_attn_forward_method (:969-1039) for the Triton backend: multi-head form for a
        # prefill without prefix, absorbed form otherwise (decode, or extend over cached latent rows)
        if forward_batch.forward_mode.is_decode() or sum(forward_batch.extend_prefix_lens_cpu) > 0:
            return self.forward_absorb(positions, hidden_states, forward_batch)
        return self.forward_normal(positions, hidden_states, forward_batch)


# =============================== the decoder stack (dense layers) ===================================
class DeepseekV2MLP(LlamaMLP):
    """gate_up -> SiLU-and-mul -> down (:167-219); column- / row-parallel over the ranks (:182-198).  The same
    module as LlamaMLP, decode fusions included: SiLU-and-mul in the gate_up GEMM's epilogue, and down_proj's
    split-K sum (tensor-parallel: its sum over the ranks) left to the next layer's norm."""


class DeepseekV2DecoderLayer(nn.Module):
    """input_layernorm -> MLA -> post_attention_layernorm -> MLP with the residual carried beside the
    hidden states (:1740-1968, the dense branch)."""

    def __init__(self, config, layer_id: int, dtype, device, quant_method_factory=None):
        super().__init__()
        self.self_attn = DeepseekV2AttentionMLA(
            config.hidden_size, config.num_attention_heads, config.qk_nope_head_dim, config.qk_rope_head_dim,
            config.v_head_dim, config.q_lora_rank, config.kv_lora_rank, layer_id, dtype, device,
            rope_theta=config.rope_theta, max_position_embeddings=config.max_position_embeddings,
            rms_norm_eps=config.rms_norm_eps, rope_scaling=config.rope_scaling, quant_method_factory=quant_method_factory)
        self.mlp = DeepseekV2MLP(config.hidden_size, config.intermediate_size, dtype, device, quant_method_factory)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, forward_batch, residual):
        at = self.self_attn
        proj_a = at.fused_qkv_a_proj_with_mqa if at.q_lora_rank is not None else None
        # decode with an fp8 "a" projection: the norm also quantises its rows (a model without q_lora feeds the
        # same rows to two projections and keeps the 16-bit tensor)
        q_in = (proj_a is not None and forward_batch.forward_mode.is_decode()
                and takes_fp8_rows_of_k(proj_a, hidden_states.shape[0]))
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states, quant_fp8=True) if q_in else self.input_layernorm(hidden_states)
        elif q_in:
            hidden_states, residual = self.input_layernorm(hidden_states, residual, quant_fp8=True)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, forward_batch)
        if takes_fp8_rows_of_k(self.mlp.gate_up_proj, hidden_states.shape[0]):   # the norm also quantises for the fp8 GEMM
            hidden_states, residual = self.post_attention_layernorm(hidden_states, residual, quant_fp8=True)
        else:
            hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


class DeepseekV2Model(nn.Module):
    def __init__(self, config, dtype, device, quant_method_factory=None):
        super().__init__()
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size, dtype, device)
        self.layers = nn.ModuleList([DeepseekV2DecoderLayer(config, i, dtype, device, quant_method_factory)
                                     for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids, positions, forward_batch):
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, forward_batch, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class DeepseekV2ForCausalLM(nn.Module):
    """DeepseekV2ForCausalLM / DeepseekV3ForCausalLM (:2064-2148) restricted to dense layers."""

    def __init__(self, config, dtype=torch.bfloat16, device="cuda", quant_method_factory=None):
        super().__init__()
        hf = config.hf_config
        n_experts = getattr(hf, "n_routed_experts", None)
        if n_experts is not None and getattr(hf, "first_k_dense_replace", 0) < config.num_hidden_layers:
            raise NotImplementedError(
                "mixture-of-experts layers are outside this path: use a configuration whose layers are all dense "
                "(first_k_dense_replace >= num_hidden_layers)")
        self.config = config
        self.dtype = dtype
        # quant_method_factory (W8A8Fp8LinearMethod): every decoder-layer projection becomes an fp8 W8A8 linear,
        # quantised per output channel when the weights are loaded (w8a8_fp8.py:118-134); kv_b_proj, an fp8
        # tensor then, feeds the absorbed form through the re-quantisation of post_load_weights()
        self.model = DeepseekV2Model(config, dtype, device, quant_method_factory)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size, dtype, device)
        self.logits_processor = LogitsProcessor(config)
        for m in self.modules():
            if isinstance(m, RMSNorm):
                m.weight.data = m.weight.data.to(dtype=dtype, device=device)
        self.to(device)

    @torch.no_grad()
    def forward(self, input_ids, positions, forward_batch):
        hidden_states = self.model(input_ids, positions, forward_batch)
        return self.logits_processor(input_ids, hidden_states, self.lm_head, forward_batch)

    @torch.no_grad()
    def load_weights(self, weights: Dict[str, torch.Tensor]) -> None:
        """HuggingFace-named tensors; q_a_proj + kv_a_proj_with_mqa are stacked into
        fused_qkv_a_proj_with_mqa and gate_proj + up_proj into gate_up_proj, as the reference's loader
        does (:2330-2345, :2417-2457)."""
        def put(param, value):
            param.data.copy_(value.to(dtype=param.dtype), non_blocking=True)

        put(self.model.embed_tokens.weight, self.model.embed_tokens.shard_from_checkpoint(weights["model.embed_tokens.weight"]))
        put(self.lm_head.weight, self.lm_head.shard_from_checkpoint(weights["lm_head.weight"]))
        put(self.model.norm.weight, weights["model.norm.weight"])
        for i, layer in enumerate(self.model.layers):
            p = f"model.layers.{i}."
            at = layer.self_attn
            put(layer.input_layernorm.weight, weights[p + "input_layernorm.weight"])
            put(layer.post_attention_layernorm.weight, weights[p + "post_attention_layernorm.weight"])
            if at.q_lora_rank is not None:
                put(at.fused_qkv_a_proj_with_mqa.weight, torch.cat(
                    [weights[p + "self_attn.q_a_proj.weight"], weights[p + "self_attn.kv_a_proj_with_mqa.weight"]], 0))
                put(at.q_a_layernorm.weight, weights[p + "self_attn.q_a_layernorm.weight"])
                put(at.q_b_proj.weight, at.q_b_proj.shard_from_checkpoint(weights[p + "self_attn.q_b_proj.weight"]))
            else:
                put(at.q_proj.weight, at.q_proj.shard_from_checkpoint(weights[p + "self_attn.q_proj.weight"]))
                put(at.kv_a_proj_with_mqa.weight, weights[p + "self_attn.kv_a_proj_with_mqa.weight"])
            put(at.kv_a_layernorm.weight, weights[p + "self_attn.kv_a_layernorm.weight"])
            put(at.kv_b_proj.weight, at.kv_b_proj.shard_from_checkpoint(weights[p + "self_attn.kv_b_proj.weight"]))
            put(at.o_proj.weight, at.o_proj.shard_from_checkpoint(weights[p + "self_attn.o_proj.weight"]))
            put(layer.mlp.gate_up_proj.weight, layer.mlp.gate_up_proj.shard_from_checkpoint(
                weights[p + "mlp.gate_proj.weight"], weights[p + "mlp.up_proj.weight"]))
            put(layer.mlp.down_proj.weight, layer.mlp.down_proj.shard_from_checkpoint(weights[p + "mlp.down_proj.weight"]))
        self.post_load_weights()

    def post_load_weights(self) -> None:
        for layer in self.model.layers:
            for lin in layer.self_attn.linears() + [layer.mlp.gate_up_proj, layer.mlp.down_proj]:
                lin.quant_method.process_weights_after_loading(lin)
            gu = layer.mlp.gate_up_proj
            if isinstance(gu.quant_method, UnquantizedLinearMethod):
                pack_gate_up_for_decode(gu)     # gate and up rows interleaved: SiLU-and-mul in the decode GEMM's epilogue
            layer.self_attn.post_load_weights()

    @torch.no_grad()
    def init_random_weights(self, seed: int = 0, std: float = 0.02) -> None:
        g = torch.Generator(device=self.model.norm.weight.device).manual_seed(seed)
        for name, p in self.named_parameters():
            w = torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32)
            if "layernorm" in name or name.endswith("norm.weight"):
                p.data.copy_((1.0 + 0.1 * w).to(p.dtype))
            else:
                p.data.copy_((std * w).to(p.dtype))
        self.post_load_weights()
