"""Llama forward graph (python/sglang/srt/models/llama.py: LlamaMLP :61-98, LlamaAttention
:101-191, LlamaDecoderLayer :194-267, LlamaModel :270-354, LlamaForCausalLM :448-481) with the
same op sequence per layer:

  input_layernorm(+residual) -> qkv_proj -> split -> RoPE -> RadixAttention -> o_proj
  -> post_attention_layernorm(+residual) -> gate_up_proj -> SiluAndMul -> down_proj

One fusion is applied where it removes launches without changing results: RoPE and the KV-cache
write of the new tokens are one kernel (the backend is then called with save_kv_cache=False).
"""
import os
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from ..distributed.parallel_state import get_tensor_model_parallel_world_size
from ..layers.activation import SiluAndMul
from ..layers.layernorm import RMSNorm
from ..layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..layers.logits_processor import LogitsProcessor
from ..layers.quantization.unquant import (
    QuantizedAct,
    SKINNY_MAX_M,
    UnquantizedLinearMethod,
    decode_linear_deferred,
    gate_up_silu_mul,
    pack_for_decode,
    pack_gate_up_for_decode,
)
from ..layers.quantization.w8a8_fp8 import W8A8Fp8LinearMethod
from ..layers.quantization.w8a8_fp8 import gate_up_silu_mul_prefill as gate_up_silu_mul_prefill_fp8
from ..layers.radix_attention import RadixAttention
from ..ops_backend import get_ops
from ..layers.rotary_embedding import get_rope
from ..layers.vocab_parallel_embedding import ParallelLMHead, VocabParallelEmbedding

# Prefill: qkv projection + RoPE + KV-cache write as one launch ("auto"), or the separate projection and
# rope kernels ("off").  Below PREFILL_QKV_MIN_TILES 256 x 256 output tiles the unfused path is taken.
PREFILL_QKV_ROPE = os.environ.get("MI355_PREFILL_QKV_ROPE", "auto")
PREFILL_QKV_MIN_TILES = 192


class LlamaMLP(nn.Module):
    def __init__(self, hidden_size, intermediate_size, dtype, device, quant_method_factory=None):
        super().__init__()
        qm = quant_method_factory or (lambda: None)
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, dtype,
                                                       device, qm())
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, dtype, device, qm())
        self.act_fn = SiluAndMul()

    def forward(self, x):
        if isinstance(self.gate_up_proj.quant_method, UnquantizedLinearMethod):
            x = gate_up_silu_mul(self.gate_up_proj, self.act_fn, x)
        elif (fused := gate_up_silu_mul_prefill_fp8(self.gate_up_proj, x)) is not None:
            x = fused   # prefill: fp8 GEMM with the scales and SiLU-and-mul in its epilogue
        else:
            gate_up, _ = self.gate_up_proj(x)
            if wants_fp8_input(self.down_proj, gate_up.shape[0]) and gate_up.dim() == 2:
                # SiLU-mul and the down projection's per-token quantisation in one launch
                xq, xs = get_ops().silu_and_mul_quant_fp8(gate_up, pad_to=self.down_proj.k_padded)
                x = QuantizedAct(xq, xs, gate_up.dtype)
            else:
                x = self.act_fn(gate_up)
        deferred = decode_linear_deferred(self.down_proj, x)   # summed by the next layer's norm
        if deferred is not None:
            return deferred
        x, _ = self.down_proj(x, defer_all_reduce=True)   # TP: the next norm may take the sum over ranks too
        return x


def wants_fp8_input(layer, m: int) -> bool:
    """True when `layer` is an fp8 W8A8 projection on the decode path whose producer may hand it
    per-token-quantised rows directly (QuantizedAct) instead of a 16-bit tensor."""
    from ..layers.quantization import unquant
    return (unquant.DECODE_FUSION and isinstance(layer.quant_method, W8A8Fp8LinearMethod)
            and 1 <= m <= SKINNY_MAX_M and getattr(layer, "weight_packed", None) is not None)


def takes_fp8_rows_of_k(layer, m: int) -> bool:
    """wants_fp8_input for a producer that cannot pad its rows (the norms): the packed GEMM must stream exactly K."""
    return wants_fp8_input(layer, m) and layer.k_padded == layer.weight.shape[1]


class LlamaAttention(nn.Module):
    def __init__(self, config, layer_id: int, dtype, device, quant_method_factory=None):
        super().__init__()
        qm = quant_method_factory or (lambda: None)
        tp = get_tensor_model_parallel_world_size()
        self.hidden_size = config.hidden_size
        self.total_num_heads = config.num_attention_heads
        self.total_num_kv_heads = config.num_key_value_heads
        self.num_heads = self.total_num_heads // tp
        self.num_kv_heads = max(1, self.total_num_kv_heads // tp)
        self.head_dim = config.head_dim
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        self.qkv_proj = QKVParallelLinear(self.hidden_size, self.head_dim, self.total_num_heads,
                                          self.total_num_kv_heads, dtype, device, qm())
        self.o_proj = RowParallelLinear(self.total_num_heads * self.head_dim, self.hidden_size,
                                        dtype, device, qm())
        self.rotary_emb = get_rope(self.head_dim, rotary_dim=self.head_dim,
                                   max_position=config.max_position_embeddings,
                                   base=config.rope_theta, rope_scaling=config.rope_scaling,
                                   is_neox_style=True, dtype=dtype)
        self.attn = RadixAttention(self.num_heads, self.head_dim, self.scaling,
                                   num_kv_heads=self.num_kv_heads, layer_id=layer_id)

    def _fused_decode(self, positions, hidden_states, forward_batch):
        """Decode fast path: the qkv GEMM leaves its split-K partials and ONE attention launch does
        the reduction, RoPE, the KV-cache write and the attention.  None when it does not apply."""
        backend = forward_batch.attn_backend
        pool = forward_batch.token_to_kv_pool
        if (not forward_batch.forward_mode.is_decode()
                or getattr(pool, "dtype", None) not in (hidden_states.dtype, torch.float8_e4m3fn)
                or not backend.supports_fused_qkv(self.attn, self.rotary_emb, hidden_states.dtype)):
            return None
        part = decode_linear_deferred(self.qkv_proj, hidden_states, needs_all_reduce=False)
        if part is None:
            return None
        return backend.forward_decode_fused_qkv(part.part, positions, self.rotary_emb.cos_sin_cache, self.attn,
                                                forward_batch, part.x_scale, part.w_scale,
                                                out_dtype=hidden_states.dtype)

    def _fused_prefill_qkv(self, positions, hidden_states, forward_batch):
        """Prefill fast path: ONE launch computes the qkv projection, rotates q and k and writes the new
        k / v rows into the KV cache (the rotation and the cache write run in the GEMM's epilogue).
        Returns the qkv tensor (q, k rotated) or None when it does not apply."""
        if (PREFILL_QKV_ROPE == "off" or not isinstance(hidden_states, torch.Tensor) or not hidden_states.is_cuda
                or forward_batch.forward_mode.is_decode() or getattr(self.qkv_proj, "bias", None) is not None):
            return None
        x = hidden_states.view(-1, hidden_states.shape[-1])
        m, k = x.shape
        qm = self.qkv_proj.quant_method
        fp8 = isinstance(qm, W8A8Fp8LinearMethod)
        w = self.qkv_proj.weight
        if not fp8 and (type(qm) is not UnquantizedLinearMethod or w.dtype != x.dtype):
            return None
        if fp8 and w.dtype != torch.float8_e4m3fn:
            return None
        kb, vb = forward_batch.token_to_kv_pool.get_kv_buffer(self.attn.layer_id)
        rope = self.rotary_emb
        ops = get_ops()
        n_tiles = ((m + 255) // 256) * (w.shape[0] // 256)
        if (kb.dtype != x.dtype or n_tiles < PREFILL_QKV_MIN_TILES
                or not ops.qkv_proj_rope_tiled_supported(m, k, self.num_heads, self.num_kv_heads, self.head_dim,
                                                         rope.cos_sin_cache.shape[1], rope.is_neox_style, x.dtype, fp8)):
            return None
        xs = ws = None
        if fp8:
            x, xs = ops.per_token_quant_fp8(x)
            ws = self.qkv_proj.weight_scale.view(-1)
        return ops.qkv_proj_rope_tiled(x, w, self.num_heads, self.num_kv_heads, self.head_dim, positions.flatten(),
                                       rope.cos_sin_cache, rope.is_neox_style, kb, vb, forward_batch.out_cache_loc,
                                       x_scale=xs, w_scale=ws, out_dtype=hidden_states.dtype)

    def forward(self, positions, hidden_states, forward_batch):
        attn_output = self._fused_decode(positions, hidden_states, forward_batch)
        if attn_output is not None:
            deferred = decode_linear_deferred(self.o_proj, attn_output)
            if deferred is not None:
                return deferred
            return self.o_proj(attn_output, defer_all_reduce=True)[0]
        qkv = self._fused_prefill_qkv(positions, hidden_states, forward_batch)
        fused = qkv is not None
        if not fused:
            qkv, _ = self.qkv_proj(hidden_states)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        pool = forward_batch.token_to_kv_pool
        kb, vb = pool.get_kv_buffer(self.attn.layer_id)
        if fused:
            attn_output = self.attn(q, k, v, forward_batch, save_kv_cache=False)
        elif kb.dtype == q.dtype:
            self.rotary_emb.forward_with_kv_write(positions, q, k, v, kb, vb, forward_batch.out_cache_loc)
            attn_output = self.attn(q, k, v, forward_batch, save_kv_cache=False)
        else:   # fp8 KV cache: the attention backend's own (quantising) KV write
            q, k = self.rotary_emb(positions, q, k)
            attn_output = self.attn(q, k, v, forward_batch, save_kv_cache=True)
        deferred = decode_linear_deferred(self.o_proj, attn_output)   # summed by post_attention_layernorm
        if deferred is not None:
            return deferred
        output, _ = self.o_proj(attn_output, defer_all_reduce=True)
        return output


class LlamaDecoderLayer(nn.Module):
    def __init__(self, config, layer_id: int, dtype, device, quant_method_factory=None):
        super().__init__()
        self.self_attn = LlamaAttention(config, layer_id, dtype, device, quant_method_factory)
        self.mlp = LlamaMLP(config.hidden_size, config.intermediate_size, dtype, device,
                            quant_method_factory)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, forward_batch,
                residual: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        m = hidden_states.shape[0]
        q_in = takes_fp8_rows_of_k(self.self_attn.qkv_proj, m)      # norm also quantises for the fp8 GEMM behind it
        q_mlp = takes_fp8_rows_of_k(self.mlp.gate_up_proj, m)
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states, quant_fp8=True) if q_in else self.input_layernorm(hidden_states)
        elif q_in:
            hidden_states, residual = self.input_layernorm(hidden_states, residual, quant_fp8=True)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, forward_batch)
        if q_mlp:
            hidden_states, residual = self.post_attention_layernorm(hidden_states, residual, quant_fp8=True)
        else:
            hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class LlamaModel(nn.Module):
    def __init__(self, config, dtype, device, quant_method_factory=None):
        super().__init__()
        self.config = config
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size, dtype, device)
        self.layers = nn.ModuleList(
            [LlamaDecoderLayer(config, i, dtype, device, quant_method_factory)
             for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids, positions, forward_batch):
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, forward_batch, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class LlamaForCausalLM(nn.Module):
    def __init__(self, config, dtype=torch.bfloat16, device="cuda", quant_method_factory=None):
        super().__init__()
        self.config = config
        self.dtype = dtype
        self.model = LlamaModel(config, dtype, device, quant_method_factory)
        if config.tie_word_embeddings:
            self.lm_head = self.model.embed_tokens
        else:
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
        """`weights`: HuggingFace-named full tensors (q_proj/k_proj/v_proj, gate_proj/up_proj are
        stacked into qkv_proj / gate_up_proj as the reference's stacked_params_mapping does,
        models/llama.py:428-435); each rank keeps its TP shard."""
        def put(param, value):
            param.data.copy_(value.to(dtype=param.dtype), non_blocking=True)

        put(self.model.embed_tokens.weight, self.model.embed_tokens.shard_from_checkpoint(weights["model.embed_tokens.weight"]))
        if not self.config.tie_word_embeddings:
            put(self.lm_head.weight, self.lm_head.shard_from_checkpoint(weights["lm_head.weight"]))
        put(self.model.norm.weight, weights["model.norm.weight"])
        for i, layer in enumerate(self.model.layers):
            p = f"model.layers.{i}."
            put(layer.input_layernorm.weight, weights[p + "input_layernorm.weight"])
            put(layer.post_attention_layernorm.weight, weights[p + "post_attention_layernorm.weight"])
            at, mlp = layer.self_attn, layer.mlp
            put(at.qkv_proj.weight, at.qkv_proj.shard_from_checkpoint(
                weights[p + "self_attn.q_proj.weight"], weights[p + "self_attn.k_proj.weight"],
                weights[p + "self_attn.v_proj.weight"]))
            put(at.o_proj.weight, at.o_proj.shard_from_checkpoint(weights[p + "self_attn.o_proj.weight"]))
            put(mlp.gate_up_proj.weight, mlp.gate_up_proj.shard_from_checkpoint(
                weights[p + "mlp.gate_proj.weight"], weights[p + "mlp.up_proj.weight"]))
            put(mlp.down_proj.weight, mlp.down_proj.shard_from_checkpoint(weights[p + "mlp.down_proj.weight"]))
        self.process_weights_after_loading()

    def process_weights_after_loading(self):
        for layer in self.model.layers:
            for lin in (layer.self_attn.qkv_proj, layer.self_attn.o_proj, layer.mlp.gate_up_proj,
                        layer.mlp.down_proj):
                lin.quant_method.process_weights_after_loading(lin)
            gu = layer.mlp.gate_up_proj
            if isinstance(gu.quant_method, UnquantizedLinearMethod):
                pack_gate_up_for_decode(gu)
        pack_for_decode(self.lm_head)

    @torch.no_grad()
    def init_random_weights(self, seed: int = 0, std: float = 0.02) -> None:
        """Fill every parameter on the device with seeded N(0, std) (norms 1 + 0.1 N(0,1)):
        a synthetic checkpoint of the right shapes for benchmarking without host traffic."""
        g = torch.Generator(device=self.model.norm.weight.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if p.dtype == torch.float8_e4m3fn:
                continue
            w = torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32)
            if "layernorm" in name or name.endswith("norm.weight"):
                p.data.copy_((1.0 + 0.1 * w).to(p.dtype))
            else:
                p.data.copy_((std * w).to(p.dtype))
        self.process_weights_after_loading()
