"""Rotary position embedding from a cos|sin cache.

Interface of the reference's RotaryEmbedding / Llama3RotaryEmbedding / get_rope
(python/sglang/srt/layers/rotary_embedding.py:78-250, 783-827, 1178-1395):
`forward(positions, query, key, offsets=None) -> (query, key)`, rotating IN PLACE like the
reference's forward_cuda/forward_hip (:226-244).  The cache is [max_pos, rotary_dim] =
cat(cos, sin) and, as on the reference's non-CUDA platforms (:98-101), is stored in the model
dtype.
"""
import math
from typing import Any, Dict, Optional, Tuple

import torch

from ..custom_op import CustomOp
from ..ops_backend import get_ops


def rope_inverse_frequencies(rotary_dim: int, base: float,
                             llama3: Optional[Dict[str, float]] = None) -> torch.Tensor:
    """theta_i = base^(-2i/rotary_dim); with Llama-3 scaling, long wavelengths are divided by
    `factor`, short ones kept, and the band in between blended linearly
    (reference: Llama3RotaryEmbedding._compute_inv_freq, rotary_embedding.py:806-827)."""
    exponent = torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim
    inv_freq = 1.0 / (base ** exponent)
    if llama3 is None:
        return inv_freq
    factor = llama3["factor"]
    lo, hi = llama3["low_freq_factor"], llama3["high_freq_factor"]
    orig = llama3["original_max_position_embeddings"]
    wavelen = 2 * math.pi / inv_freq
    blend = (orig / wavelen - lo) / (hi - lo) if lo != hi else 0
    scaled = inv_freq / factor
    mid = (1 - blend) * scaled + blend * inv_freq
    return torch.where(wavelen < orig / hi, inv_freq, torch.where(wavelen > orig / lo, scaled, mid))


def build_cos_sin_cache(rotary_dim: int, max_position: int, base: float,
                        llama3: Optional[Dict[str, float]] = None) -> torch.Tensor:
    inv_freq = rope_inverse_frequencies(rotary_dim, base, llama3)
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


def yarn_get_mscale(scale: float = 1, mscale: float = 1) -> float:
    """Magnitude correction of YaRN-interpolated attention (rotary_embedding.py:635-638)."""
    return 1.0 if scale <= 1 else 0.1 * mscale * math.log(scale) + 1.0


def yarn_inverse_frequencies(rotary_dim: int, base: float, factor: float, original_max_position: int,
                             beta_fast: float = 32, beta_slow: float = 1,
                             extrapolation_factor: float = 1) -> torch.Tensor:
    """YaRN: feature pairs that turn more than beta_fast times over the original context keep their
    frequency, those that turn fewer than beta_slow times are interpolated (divided by `factor`), the
    pairs in between blend linearly (DeepseekScalingRotaryEmbedding._compute_inv_freq,
    rotary_embedding.py:382-418, 685-711)."""
    half = rotary_dim // 2
    wavelength_base = base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim)

    def pair_index_for(rotations):   # the (fractional) pair whose wave turns `rotations` times in the original context
        return rotary_dim * math.log(original_max_position / (rotations * 2 * math.pi)) / (2 * math.log(base))

    lo = max(math.floor(pair_index_for(beta_fast)), 0)
    hi = min(math.ceil(pair_index_for(beta_slow)), rotary_dim - 1)
    if lo == hi:
        hi += 0.001
    ramp = torch.clamp((torch.arange(half, dtype=torch.float) - lo) / (hi - lo), 0, 1)
    keep = (1 - ramp) * extrapolation_factor
    return (1.0 / (factor * wavelength_base)) * (1 - keep) + (1.0 / wavelength_base) * keep


class RotaryEmbedding(CustomOp):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 is_neox_style: bool, dtype: torch.dtype,
                 llama3_scaling: Optional[Dict[str, float]] = None) -> None:
        super().__init__()
        self.head_size = head_size
        self.rotary_dim = rotary_dim
        self.max_position_embeddings = max_position_embeddings
        self.base = base
        self.is_neox_style = is_neox_style
        self.dtype = dtype
        cache = self._compute_cos_sin_cache(llama3_scaling)
        self.register_buffer("cos_sin_cache", cache.to(dtype), persistent=False)

    def _compute_cos_sin_cache(self, llama3_scaling=None) -> torch.Tensor:
        return build_cos_sin_cache(self.rotary_dim, self.max_position_embeddings, self.base, llama3_scaling)

    def forward_hip(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                    offsets: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if offsets is not None:
            positions = positions + offsets
        get_ops().rope_inplace(positions.flatten(), query, key, self.head_size,
                               self.cos_sin_cache, self.is_neox_style)
        return query, key

    def forward_with_kv_write(self, positions, query, key, value, k_buffer, v_buffer, loc):
        """RoPE fused with the KV-cache write of the new tokens (saves the two index_put
        launches of MHATokenToKVPool.set_kv_buffer, memory_pool.py:457-458)."""
        get_ops().rope_inplace(positions.flatten(), query, key, self.head_size,
                               self.cos_sin_cache, self.is_neox_style, v=value,
                               k_buffer=k_buffer, v_buffer=v_buffer, loc=loc)
        return query, key


class Llama3RotaryEmbedding(RotaryEmbedding):
    def __init__(self, head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype,
                 scaling_factor, low_freq_factor, high_freq_factor, orig_max_position) -> None:
        super().__init__(head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype,
                         llama3_scaling=dict(factor=scaling_factor, low_freq_factor=low_freq_factor,
                                             high_freq_factor=high_freq_factor,
                                             original_max_position_embeddings=orig_max_position))


class DeepseekScalingRotaryEmbedding(RotaryEmbedding):
    """YaRN-scaled table for DeepSeek-V2/V3 (rotary_embedding.py:641-765): interpolated inverse
    frequencies, original_max_position * factor rows, cos / sin multiplied by
    mscale(factor, mscale) / mscale(factor, mscale_all_dim) * attn_factor before the cast to the model
    dtype.  The rotation itself is the ordinary one (the kernel reads whatever the table holds); the
    attention module multiplies its softmax scale by mscale(factor, mscale_all_dim)^2."""

    def __init__(self, head_size, rotary_dim, max_position_embeddings, base, is_neox_style, scaling_factor, dtype, *,
                 extrapolation_factor: float = 1, attn_factor: float = 1, beta_fast: float = 32, beta_slow: float = 1,
                 mscale: float = 1, mscale_all_dim: float = 0) -> None:
        self.scaling_factor = scaling_factor
        self.extrapolation_factor, self.attn_factor = extrapolation_factor, attn_factor
        self.beta_fast, self.beta_slow = beta_fast, beta_slow
        self.mscale = float(yarn_get_mscale(scaling_factor, float(mscale))
                            / yarn_get_mscale(scaling_factor, float(mscale_all_dim)) * attn_factor)
        super().__init__(head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype)

    def _compute_cos_sin_cache(self, llama3_scaling=None) -> torch.Tensor:
        inv_freq = yarn_inverse_frequencies(self.rotary_dim, self.base, self.scaling_factor, self.max_position_embeddings,
                                            self.beta_fast, self.beta_slow, self.extrapolation_factor)
        t = torch.arange(self.max_position_embeddings * self.scaling_factor, dtype=torch.float32)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        return torch.cat((freqs.cos() * self.mscale, freqs.sin() * self.mscale), dim=-1)


_ROPE_DICT: Dict[Tuple, RotaryEmbedding] = {}


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float,
             is_neox_style: bool = True, rope_scaling: Optional[Dict[str, Any]] = None,
             dtype: Optional[torch.dtype] = None, partial_rotary_factor: float = 1.0) -> RotaryEmbedding:
    """Same arguments and caching behaviour as the reference's get_rope
    (rotary_embedding.py:1178-1395); scaling types beyond None / "default" / "llama3" /
    "deepseek_yarn" belong to model families outside this package."""
    if dtype is None:
        dtype = torch.get_default_dtype()
    if partial_rotary_factor < 1.0:
        rotary_dim = int(rotary_dim * partial_rotary_factor)
    scaling_key = None
    if rope_scaling is not None:
        scaling_key = tuple(sorted((k, tuple(v) if isinstance(v, list) else v)
                                   for k, v in rope_scaling.items()))
    key = (head_size, rotary_dim, max_position, base, is_neox_style, scaling_key, dtype)
    if key in _ROPE_DICT:
        return _ROPE_DICT[key]
    kind = None
    if rope_scaling is not None:
        kind = rope_scaling.get("rope_type", rope_scaling.get("type"))
        if kind is None:
            raise ValueError("Unknown RoPE scaling type")
    if kind in (None, "default"):
        rope = RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype)
    elif kind == "llama3":
        rope = Llama3RotaryEmbedding(
            head_size, rotary_dim, max_position, base, is_neox_style, dtype,
            rope_scaling["factor"], rope_scaling["low_freq_factor"],
            rope_scaling["high_freq_factor"], rope_scaling["original_max_position_embeddings"])
    elif kind == "deepseek_yarn":
        extra = {k: v for k, v in rope_scaling.items()
                 if k in ("extrapolation_factor", "attn_factor", "beta_fast", "beta_slow", "mscale", "mscale_all_dim")}
        rope = DeepseekScalingRotaryEmbedding(
            head_size, rotary_dim, rope_scaling["original_max_position_embeddings"], base, is_neox_style,
            rope_scaling["factor"], dtype, **extra)
    else:
        raise ValueError(f"RoPE scaling type {kind!r} is not supported by this backend")
    _ROPE_DICT[key] = rope
    return rope
