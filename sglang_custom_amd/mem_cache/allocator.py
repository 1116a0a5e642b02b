"""Token-slot allocator for page_size = 1
(python/sglang/srt/mem_cache/allocator.py:117-160): slots 1..size are handed out in order from
`free_pages`; slot 0 is never allocated (padding sink); freed slots collect in `release_pages`
and are merged + sorted only when the free list runs short."""
from typing import Optional

import torch


class TokenToKVPoolAllocator:
    def __init__(self, size: int, dtype: torch.dtype, device: str, kvcache):
        self.size = size
        self.page_size = 1
        self.dtype = dtype
        self.device = device
        self._kvcache = kvcache
        self.clear()

    def clear(self):
        self.free_pages = torch.arange(1, self.size + 1, dtype=torch.int64, device=self.device)
        self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)

    def get_kvcache(self):
        return self._kvcache

    def available_size(self) -> int:
        return len(self.free_pages) + len(self.release_pages)

    def merge_and_sort_free(self):
        if len(self.release_pages) > 0:
            self.free_pages, _ = torch.sort(torch.cat((self.free_pages, self.release_pages)))
            self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        if need_size > len(self.free_pages):
            self.merge_and_sort_free()
        if need_size > len(self.free_pages):
            return None
        select_index = self.free_pages[:need_size]
        self.free_pages = self.free_pages[need_size:]
        return select_index

    def free(self, free_index: torch.Tensor):
        if free_index.numel() == 0:
            return
        self.release_pages = torch.cat((self.release_pages, free_index))
