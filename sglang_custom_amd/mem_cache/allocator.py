"""Token-slot allocators.  page_size = 1
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


class PagedTokenToKVPoolAllocator:
    """Slot allocator for page_size > 1 (python/sglang/srt/mem_cache/allocator.py:404-570): the
    slots of one request are page-aligned -- whole pages, plus the tail of its last page.  Pages
    1..num_pages are handed out from `free_pages` in order (page 0 holds the padding sink); a free
    returns every page the given slots touch.  alloc_extend / alloc_decode compute the slot list on
    the device (ops.alloc_extend_pages / alloc_decode_pages)."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, device: str, kvcache):
        assert page_size > 1
        self.size = size
        self.page_size = page_size
        self.num_pages = size // page_size
        self.dtype = dtype
        self.device = device
        self._kvcache = kvcache
        self.clear()

    def clear(self):
        self.free_pages = torch.arange(1, self.num_pages + 1, dtype=torch.int64, device=self.device)
        self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)

    def get_kvcache(self):
        return self._kvcache

    def available_size(self) -> int:
        return (len(self.free_pages) + len(self.release_pages)) * self.page_size

    def merge_and_sort_free(self):
        if len(self.release_pages) > 0:
            self.free_pages, _ = torch.sort(torch.cat((self.free_pages, self.release_pages)))
            self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        """Whole pages only (need_size % page_size == 0)."""
        assert need_size % self.page_size == 0
        num_pages = need_size // self.page_size
        if num_pages > len(self.free_pages):
            self.merge_and_sort_free()
        if num_pages > len(self.free_pages):
            return None
        out_pages = self.free_pages[:num_pages]
        self.free_pages = self.free_pages[num_pages:]
        return (out_pages[:, None] * self.page_size
                + torch.arange(self.page_size, device=self.device, dtype=torch.int64)).reshape(-1)

    def _ensure(self, num_new_pages: int) -> bool:
        if num_new_pages > len(self.free_pages):
            self.merge_and_sort_free()
        return num_new_pages <= len(self.free_pages)

    def alloc_extend(self, prefix_lens: torch.Tensor, seq_lens: torch.Tensor, last_loc: torch.Tensor,
                     extend_num_tokens: int, prefix_lens_cpu=None, seq_lens_cpu=None) -> Optional[torch.Tensor]:
        from ..ops_backend import get_ops

        ps = self.page_size
        pre = prefix_lens_cpu if prefix_lens_cpu is not None else prefix_lens.tolist()
        seq = seq_lens_cpu if seq_lens_cpu is not None else seq_lens.tolist()
        need = sum((s + ps - 1) // ps - (p + ps - 1) // ps for p, s in zip(pre, seq))
        if not self._ensure(need):
            return None
        out, used = get_ops().alloc_extend_pages(prefix_lens, seq_lens, last_loc, self.free_pages, ps,
                                                 extend_num_tokens)
        assert used == need
        self.free_pages = self.free_pages[used:]
        return out.to(self.device)

    def alloc_decode(self, seq_lens: torch.Tensor, last_loc: torch.Tensor, seq_lens_cpu=None) -> Optional[torch.Tensor]:
        from ..ops_backend import get_ops

        ps = self.page_size
        seq = seq_lens_cpu if seq_lens_cpu is not None else seq_lens.tolist()
        need = sum(1 for s in seq if (s - 1) % ps == 0)
        if not self._ensure(need):
            return None
        out, used = get_ops().alloc_decode_pages(seq_lens, last_loc, self.free_pages, ps)
        assert used == need
        self.free_pages = self.free_pages[used:]
        return out.to(self.device)

    def free(self, free_index: torch.Tensor):
        if free_index.numel() == 0:
            return
        pages = torch.unique(free_index.to(torch.int64) // self.page_size)
        self.release_pages = torch.cat((pages, self.release_pages))

