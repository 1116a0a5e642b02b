"""ModelConfig: head / KV geometry the hot path needs (the fields of the reference's
python/sglang/srt/configs/model_config.py that model_runner, the pools and the attention
backend read), built from a HuggingFace-style config dict or config.json."""
import json
import os
from types import SimpleNamespace
from typing import Optional

import torch

_DTYPES = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16,
           "half": torch.float16, "auto": None}


class ModelConfig:
    def __init__(self, hf_config: dict, dtype: str = "auto", context_length: Optional[int] = None,
                 quantization: Optional[str] = None):
        c = dict(hf_config)
        self.hf_config = SimpleNamespace(**c)
        self.hidden_size = c["hidden_size"]
        self.num_hidden_layers = c["num_hidden_layers"]
        self.num_attention_heads = c["num_attention_heads"]
        self.num_key_value_heads = c.get("num_key_value_heads", self.num_attention_heads)
        self.head_dim = c.get("head_dim") or self.hidden_size // self.num_attention_heads
        self.v_head_dim = self.head_dim
        self.intermediate_size = c["intermediate_size"]
        self.vocab_size = c["vocab_size"]
        self.rms_norm_eps = c.get("rms_norm_eps", 1e-6)
        self.rope_theta = c.get("rope_theta", 10000.0)
        self.rope_scaling = c.get("rope_scaling")
        self.max_position_embeddings = c.get("max_position_embeddings", 8192)
        self.tie_word_embeddings = c.get("tie_word_embeddings", False)
        self.context_len = context_length or self.max_position_embeddings
        self.quantization = quantization
        want = _DTYPES[dtype] if dtype in _DTYPES else None
        if want is None:
            want = _DTYPES.get(str(c.get("torch_dtype", "bfloat16")), torch.bfloat16)
        self.dtype = want
        self.is_encoder_decoder = False
        # multi-head latent attention (model_config.py:186-204): the cache holds one latent row per token
        arch = (c.get("architectures") or [""])[0]
        self.is_mla = arch in ("DeepseekV2ForCausalLM", "DeepseekV3ForCausalLM")
        if self.is_mla:
            self.head_dim = 256
            self.kv_lora_rank = c["kv_lora_rank"]
            self.qk_nope_head_dim = c["qk_nope_head_dim"]
            self.qk_rope_head_dim = c["qk_rope_head_dim"]
            self.v_head_dim = c["v_head_dim"]
            self.q_lora_rank = c.get("q_lora_rank")
            self.num_key_value_heads = 1

    @classmethod
    def from_path(cls, model_path: str, **kw) -> "ModelConfig":
        with open(os.path.join(model_path, "config.json")) as f:
            return cls(json.load(f), **kw)

    def get_num_kv_heads(self, tp_size: int) -> int:
        return max(1, self.num_key_value_heads // tp_size)
