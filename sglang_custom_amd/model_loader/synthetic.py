"""Synthetic Llama checkpoints (no network, no real weights in this environment).

The reference's ``--load-format dummy`` initialiser draws uniform(-1e-3, 1e-3)
(python/sglang/srt/model_loader/weight_utils.py:750-779), which gives logits of ~1e-6 and
is useless for arg-max parity.  This module instead draws seeded N(0, std) projection
weights and (1 + 0.1 N(0,1)) norm weights, named exactly like a HuggingFace Llama
checkpoint (q_proj/k_proj/v_proj/o_proj/gate_proj/up_proj/down_proj), so the same tensors
can be written to safetensors for the reference's loader and fed to this package's loader.

Generation is on the CPU with a private torch.Generator, so the same seed gives the same
bits wherever this image runs.
"""
from collections import OrderedDict
from typing import Dict, Tuple

import torch

# Named shape presets (public model cards).  "tiny*" presets are for parity tests.
LLAMA_PRESETS = {
    "tiny-llama-d64": dict(
        hidden_size=512, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2,
        head_dim=64, intermediate_size=1024, vocab_size=2048, rms_norm_eps=1e-5,
        rope_theta=500000.0, max_position_embeddings=2048, tie_word_embeddings=False,
        rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0,
                          high_freq_factor=4.0, original_max_position_embeddings=256),
    ),
    "tiny-llama-d128": dict(
        hidden_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=1,
        head_dim=128, intermediate_size=1536, vocab_size=4096, rms_norm_eps=1e-5,
        rope_theta=500000.0, max_position_embeddings=2048, tie_word_embeddings=True,
        rope_scaling=None,
    ),
    "llama-3.2-1b": dict(
        hidden_size=2048, num_hidden_layers=16, num_attention_heads=32, num_key_value_heads=8,
        head_dim=64, intermediate_size=8192, vocab_size=128256, rms_norm_eps=1e-5,
        rope_theta=500000.0, max_position_embeddings=131072, tie_word_embeddings=True,
        rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0,
                          high_freq_factor=4.0, original_max_position_embeddings=8192),
    ),
    "llama-3-8b": dict(
        hidden_size=4096, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
        head_dim=128, intermediate_size=14336, vocab_size=128256, rms_norm_eps=1e-5,
        rope_theta=500000.0, max_position_embeddings=8192, tie_word_embeddings=False,
        rope_scaling=None,
    ),
    "llama-3-70b": dict(
        hidden_size=8192, num_hidden_layers=80, num_attention_heads=64, num_key_value_heads=8,
        head_dim=128, intermediate_size=28672, vocab_size=128256, rms_norm_eps=1e-5,
        rope_theta=500000.0, max_position_embeddings=8192, tie_word_embeddings=False,
        rope_scaling=None,
    ),
}


# DeepSeek (multi-head latent attention) presets whose layers are all dense.  "tiny-deepseek*": the
# attention geometry of one DeepSeek-V3 tensor-parallel rank of 8 (16 heads, 128 + 64 query/key
# features, 128 value features, latent rank 512) on a small hidden size; "-yarn" adds DeepSeek-V3's
# published rope_scaling.
_YARN_V3 = dict(type="yarn", factor=40.0, original_max_position_embeddings=4096, beta_fast=32, beta_slow=1,
                mscale=1.0, mscale_all_dim=1.0)
DEEPSEEK_PRESETS = {
    "tiny-deepseek": dict(
        hidden_size=512, num_hidden_layers=2, num_attention_heads=16, q_lora_rank=None, kv_lora_rank=512,
        qk_nope_head_dim=128, qk_rope_head_dim=64, v_head_dim=128, intermediate_size=1024, vocab_size=2048,
        rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=4096, rope_scaling=None),
    # DeepSeek-V3's own attention / dense-MLP geometry (hidden 7168, 128 heads, q_lora 1536, latent 512 + 64,
    # dense intermediate 18432 as in its first three layers, vocabulary 129280) with FOUR dense layers: the
    # per-layer decode cost of the dense part at the real shapes (the 58 mixture-of-experts layers are not built)
    "deepseek-v3-dense4": dict(
        hidden_size=7168, num_hidden_layers=4, num_attention_heads=128, q_lora_rank=1536, kv_lora_rank=512,
        qk_nope_head_dim=128, qk_rope_head_dim=64, v_head_dim=128, intermediate_size=18432, vocab_size=129280,
        rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=8192, rope_scaling=_YARN_V3),
    # ONE dense layer of DeepSeek-V3 at its real attention / MLP shapes (all 128 heads) under a small vocabulary:
    # small enough for a CPU oracle, for end-to-end checks at the real shapes
    "deepseek-v3-dense1-vocab2k": dict(
        hidden_size=7168, num_hidden_layers=1, num_attention_heads=128, q_lora_rank=1536, kv_lora_rank=512,
        qk_nope_head_dim=128, qk_rope_head_dim=64, v_head_dim=128, intermediate_size=18432, vocab_size=2048,
        rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=8192, rope_scaling=_YARN_V3),
    "tiny-deepseek-yarn": dict(
        hidden_size=512, num_hidden_layers=2, num_attention_heads=16, q_lora_rank=192, kv_lora_rank=512,
        qk_nope_head_dim=128, qk_rope_head_dim=64, v_head_dim=128, intermediate_size=1024, vocab_size=2048,
        rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=4096, rope_scaling=_YARN_V3),
}


def hf_config_dict(preset: str, torch_dtype: str = "bfloat16") -> dict:
    """A HuggingFace-style config.json body for the preset (what the reference's
    ModelConfig.from_server_args reads, python/sglang/srt/configs/model_config.py)."""
    if preset in DEEPSEEK_PRESETS:
        c = dict(DEEPSEEK_PRESETS[preset])
        c.update(architectures=["DeepseekV3ForCausalLM"], model_type="deepseek_v3", hidden_act="silu",
                 n_routed_experts=None, first_k_dense_replace=c["num_hidden_layers"], tie_word_embeddings=False,
                 bos_token_id=0, eos_token_id=1, torch_dtype=torch_dtype, initializer_range=0.02, use_cache=True)
        return c
    c = dict(LLAMA_PRESETS[preset])
    c.update(
        architectures=["LlamaForCausalLM"], model_type="llama", hidden_act="silu",
        attention_bias=False, mlp_bias=False, bos_token_id=1, eos_token_id=2,
        torch_dtype=torch_dtype, initializer_range=0.02, use_cache=True,
    )
    return c


def llama_param_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    h = cfg["hidden_size"]
    d = cfg.get("head_dim") or h // cfg["num_attention_heads"]
    hq, hkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    inter, vocab = cfg["intermediate_size"], cfg["vocab_size"]
    shapes = OrderedDict()
    shapes["model.embed_tokens.weight"] = (vocab, h)
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        shapes[p + "input_layernorm.weight"] = (h,)
        shapes[p + "self_attn.q_proj.weight"] = (hq * d, h)
        shapes[p + "self_attn.k_proj.weight"] = (hkv * d, h)
        shapes[p + "self_attn.v_proj.weight"] = (hkv * d, h)
        shapes[p + "self_attn.o_proj.weight"] = (h, hq * d)
        shapes[p + "post_attention_layernorm.weight"] = (h,)
        shapes[p + "mlp.gate_proj.weight"] = (inter, h)
        shapes[p + "mlp.up_proj.weight"] = (inter, h)
        shapes[p + "mlp.down_proj.weight"] = (h, inter)
    shapes["model.norm.weight"] = (h,)
    if not cfg.get("tie_word_embeddings", False):
        shapes["lm_head.weight"] = (vocab, h)
    return shapes


def deepseek_param_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    """HuggingFace names of a dense-layer DeepSeek-V2/V3 checkpoint."""
    h, H = cfg["hidden_size"], cfg["num_attention_heads"]
    nope, rope, v, kr, qr = (cfg["qk_nope_head_dim"], cfg["qk_rope_head_dim"], cfg["v_head_dim"], cfg["kv_lora_rank"],
                             cfg.get("q_lora_rank"))
    inter, vocab = cfg["intermediate_size"], cfg["vocab_size"]
    shapes = OrderedDict()
    shapes["model.embed_tokens.weight"] = (vocab, h)
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        shapes[p + "input_layernorm.weight"] = (h,)
        if qr is not None:
            shapes[p + "self_attn.q_a_proj.weight"] = (qr, h)
            shapes[p + "self_attn.q_a_layernorm.weight"] = (qr,)
            shapes[p + "self_attn.q_b_proj.weight"] = (H * (nope + rope), qr)
        else:
            shapes[p + "self_attn.q_proj.weight"] = (H * (nope + rope), h)
        shapes[p + "self_attn.kv_a_proj_with_mqa.weight"] = (kr + rope, h)
        shapes[p + "self_attn.kv_a_layernorm.weight"] = (kr,)
        shapes[p + "self_attn.kv_b_proj.weight"] = (H * (nope + v), kr)
        shapes[p + "self_attn.o_proj.weight"] = (h, H * v)
        shapes[p + "post_attention_layernorm.weight"] = (h,)
        shapes[p + "mlp.gate_proj.weight"] = (inter, h)
        shapes[p + "mlp.up_proj.weight"] = (inter, h)
        shapes[p + "mlp.down_proj.weight"] = (h, inter)
    shapes["model.norm.weight"] = (h,)
    shapes["lm_head.weight"] = (vocab, h)
    return shapes


def make_synthetic_llama_weights(
    cfg: dict, seed: int = 1234, std: float = 0.02, dtype: torch.dtype = torch.bfloat16,
    device: str = "cpu",
) -> Dict[str, torch.Tensor]:
    """Seeded synthetic checkpoint.  device='cpu' is bit-reproducible (used by parity
    tests and the golden generator); any other device draws with that device's generator
    (used by the benchmark, where only shapes matter)."""
    out = OrderedDict()
    if device == "cpu":
        g = torch.Generator(device="cpu").manual_seed(seed)
    else:
        g = torch.Generator(device=device).manual_seed(seed)
    is_mla = (cfg.get("architectures") or [""])[0].startswith("Deepseek")
    for name, shape in (deepseek_param_shapes(cfg) if is_mla else llama_param_shapes(cfg)).items():
        if name.endswith("layernorm.weight") or name == "model.norm.weight":
            w = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32, device=device)
        else:
            w = std * torch.randn(shape, generator=g, dtype=torch.float32, device=device)
        out[name] = w.to(dtype)
    return out
