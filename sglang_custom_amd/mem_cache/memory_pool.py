"""KV storage in HBM, same layout contract as the reference
(python/sglang/srt/mem_cache/memory_pool.py:51-98 ReqToTokenPool, :178-468 MHATokenToKVPool):

  req_to_token  int32 [max_reqs, max_context_len]   slot of token j of request r
  k_buffer[l]   T     [size + page_size, Hkv, D]    row 0 is a sink for padded tokens
  v_buffer[l]   T     [size + page_size, Hkv, Dv]

One contiguous allocation backs all layers of K (and one of V), so a layer's buffer is a view;
with 288 GB of HBM3E the pool is sized by the caller, not squeezed.  With head_major=True the
memory behind a layer is [Hkv, size + page_size, 2, D] (keys | values; two [Hkv, size + page_size, D]
allocations when D != Dv) and k_buffer[l] / v_buffer[l] are permute(1, 0, 2) views of its halves: the same tensor to every reader, but the rows one kv head's decode workgroup streams (one
head, consecutive slots) are adjacent in HBM instead of Hkv * D elements apart."""
from typing import List, Optional, Tuple, Union

import torch

from ..ops_backend import get_ops


class ReqToTokenPool:
    def __init__(self, size: int, max_context_len: int, device: str):
        self.size = size
        self.max_context_len = max_context_len
        self.device = device
        self.req_to_token = torch.zeros((size, max_context_len), dtype=torch.int32, device=device)
        self.free_slots = list(range(size))

    def write(self, indices, values):
        self.req_to_token[indices] = values

    def available_size(self) -> int:
        return len(self.free_slots)

    def alloc(self, need_size: int) -> Optional[List[int]]:
        if need_size > len(self.free_slots):
            return None
        select_index = self.free_slots[:need_size]
        self.free_slots = self.free_slots[need_size:]
        return select_index

    def free(self, free_index: Union[int, List[int]]):
        if isinstance(free_index, int):
            self.free_slots.append(free_index)
        else:
            self.free_slots.extend(free_index)

    def clear(self):
        self.free_slots = list(range(self.size))


class MHATokenToKVPool:
    def __init__(self, size: int, page_size: int, dtype: torch.dtype, head_num: int, head_dim: int,
                 layer_num: int, device: str, v_head_dim: Optional[int] = None, head_major: bool = False):
        self.size = size
        self.head_major = head_major
        self.page_size = page_size
        self.dtype = dtype
        self.store_dtype = dtype
        self.head_num = head_num
        self.head_dim = head_dim
        self.v_head_dim = v_head_dim or head_dim
        self.layer_num = layer_num
        self.device = device
        rows = size + page_size
        if head_major:
            # The head stride is rows * row_bytes.  Measured on MI355X (Llama-3-8B, 64 sequences whose
            # prompts lie 512 KiB apart in a head's region): when the 32-KiB digit of that stride is 9 or 10 (mod 16)
            # the eight heads of a sequence meet on the same DRAM channels and decode attention loses the whole gain
            # of the layout (step +2.5 %); every other digit measured is fine.  A few unused rows at the end of a
            # head's region move the stride off those digits (the allocator never hands them out).
            row_bytes = (2 if self.v_head_dim == head_dim else 1) * head_dim * torch.empty((), dtype=dtype).element_size()
            while ((rows * row_bytes >> 15) & 15) in (9, 10):
                rows += max(1, (1 << 15) // row_bytes)
        if head_major and self.v_head_dim == head_dim:
            # keys and values of a (head, slot) side by side: one 2 D run per token for the workgroup that streams them
            self._kv_all = torch.zeros((layer_num, head_num, rows, 2, head_dim), dtype=dtype, device=device)
            self._k_all, self._v_all = self._kv_all[..., 0, :], self._kv_all[..., 1, :]
            self.k_buffer = [self._k_all[i].permute(1, 0, 2) for i in range(layer_num)]
            self.v_buffer = [self._v_all[i].permute(1, 0, 2) for i in range(layer_num)]
        elif head_major:
            self._k_all = torch.zeros((layer_num, head_num, rows, head_dim), dtype=dtype, device=device)
            self._v_all = torch.zeros((layer_num, head_num, rows, self.v_head_dim), dtype=dtype, device=device)
            self.k_buffer = [self._k_all[i].permute(1, 0, 2) for i in range(layer_num)]
            self.v_buffer = [self._v_all[i].permute(1, 0, 2) for i in range(layer_num)]
        else:
            self._k_all = torch.zeros((layer_num, rows, head_num, head_dim), dtype=dtype, device=device)
            self._v_all = torch.zeros((layer_num, rows, head_num, self.v_head_dim), dtype=dtype, device=device)
            self.k_buffer = [self._k_all[i] for i in range(layer_num)]
            self.v_buffer = [self._v_all[i] for i in range(layer_num)]

    def get_kv_size_bytes(self) -> Tuple[int, int]:
        return (self._k_all.numel() * self._k_all.element_size(),
                self._v_all.numel() * self._v_all.element_size())

    def get_key_buffer(self, layer_id: int) -> torch.Tensor:
        return self.k_buffer[layer_id]

    def get_value_buffer(self, layer_id: int) -> torch.Tensor:
        return self.v_buffer[layer_id]

    def get_kv_buffer(self, layer_id: int):
        return self.k_buffer[layer_id], self.v_buffer[layer_id]

    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor,
                      k_scale=None, v_scale=None, layer_id_override: Optional[int] = None):
        layer_id = layer_id_override if layer_id_override is not None else layer.layer_id
        if self.dtype == torch.float8_e4m3fn:
            # fp8 cache (memory_pool.py:436-446): optional per-layer scales divide first, the cast to
            # e4m3fn happens inside the scatter kernel (saturating at +-448)
            if k_scale is not None:
                cache_k = cache_k / k_scale
            if v_scale is not None:
                cache_v = cache_v / v_scale
        elif cache_k.dtype != self.dtype:
            cache_k = cache_k.to(self.dtype)
            cache_v = cache_v.to(self.dtype)
        get_ops().set_kv_buffer(self.k_buffer[layer_id], self.v_buffer[layer_id], loc, cache_k, cache_v)


class MLATokenToKVPool:
    """Compressed KV cache of multi-head latent attention
    (python/sglang/srt/mem_cache/memory_pool.py:769-970): one row of kv_lora_rank + qk_rope_head_dim
    features per token per layer, [size + page_size, 1, rank + rope]; slot 0 is the sink row.  The
    value "buffer" is the first kv_lora_rank features of the same rows (:863-871) -- a view, never
    a copy, which is what lets the attention kernel read each row once for both K and V."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, kv_lora_rank: int,
                 qk_rope_head_dim: int, layer_num: int, device: str):
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        self.store_dtype = dtype
        self.kv_lora_rank = kv_lora_rank
        self.qk_rope_head_dim = qk_rope_head_dim
        self.head_num = 1
        self.head_dim = kv_lora_rank + qk_rope_head_dim
        self.v_head_dim = kv_lora_rank
        self.layer_num = layer_num
        self.device = device
        self._kv_all = torch.zeros((layer_num, size + page_size, 1, self.head_dim), dtype=dtype, device=device)
        self.kv_buffer = [self._kv_all[i] for i in range(layer_num)]

    def get_kv_size_bytes(self) -> int:
        return self._kv_all.numel() * self._kv_all.element_size()

    def get_key_buffer(self, layer_id: int) -> torch.Tensor:
        return self.kv_buffer[layer_id]

    def get_value_buffer(self, layer_id: int) -> torch.Tensor:
        return self.kv_buffer[layer_id][..., : self.kv_lora_rank]

    def get_kv_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor = None,
                      layer_id_override: Optional[int] = None):
        """cache_k [T, 1, rank + rope] is the whole latent row; cache_v is ignored (:895-909)."""
        layer_id = layer_id_override if layer_id_override is not None else layer.layer_id
        buf = self.kv_buffer[layer_id]
        # an fp8 pool (memory_pool.py:895-909 casts with cache_k.to(dtype)) is written by the scatter kernel
        # itself: round to nearest even, saturating at +-448
        row = cache_k if self.dtype == torch.float8_e4m3fn else cache_k.to(self.dtype)
        row = row.reshape(loc.numel(), 1, self.head_dim)
        # the generic scatter kernel copies a K row and a V row per token: hand it the row twice
        get_ops().set_kv_buffer(buf, buf, loc, row, row)

    def set_mla_kv_buffer(self, layer, loc: torch.Tensor, cache_k_nope: torch.Tensor, cache_k_rope: torch.Tensor):
        """Latent part and rope part arrive separately (:911-930)."""
        T = loc.numel()
        row = torch.cat([cache_k_nope.reshape(T, 1, -1), cache_k_rope.reshape(T, 1, -1)], dim=-1)
        self.set_kv_buffer(layer, loc, row)
