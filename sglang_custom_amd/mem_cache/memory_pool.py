"""KV storage in HBM, same layout contract as the reference
(python/sglang/srt/mem_cache/memory_pool.py:51-98 ReqToTokenPool, :178-468 MHATokenToKVPool):

  req_to_token  int32 [max_reqs, max_context_len]   slot of token j of request r
  k_buffer[l]   T     [size + page_size, Hkv, D]    row 0 is a sink for padded tokens
  v_buffer[l]   T     [size + page_size, Hkv, Dv]

One contiguous allocation backs all layers of K (and one of V), so a layer's buffer is a view;
with 288 GB of HBM3E the pool is sized by the caller, not squeezed."""
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
                 layer_num: int, device: str, v_head_dim: Optional[int] = None):
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        self.store_dtype = dtype
        self.head_num = head_num
        self.head_dim = head_dim
        self.v_head_dim = v_head_dim or head_dim
        self.layer_num = layer_num
        self.device = device
        rows = size + page_size
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
        if cache_k.dtype != self.dtype:
            cache_k = cache_k.to(self.dtype)
            cache_v = cache_v.to(self.dtype)
        get_ops().set_kv_buffer(self.k_buffer[layer_id], self.v_buffer[layer_id], loc, cache_k, cache_v)
