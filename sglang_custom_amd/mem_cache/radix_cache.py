"""Radix tree over token ids whose values are KV-cache slots: the prefix cache behind
"RadixAttention" (python/sglang/srt/mem_cache/radix_cache.py: TreeNode :40-80, match_prefix
:134-178, insert :180-186, cache_finished_req :188-222, cache_unfinished_req :224-270, evict
:283-306, inc/dec_lock_ref :308-338, _split_node :380-398).  With page_size > 1 keys match in
whole pages (_key_match_paged :111-120, children keyed by their first page :149): lookups and
insertions are truncated to a multiple of the page size and a request's partly filled last page
never enters the tree.

A cached prefix costs nothing on the GPU: its slots are handed to prepare_for_extend as
prefix_indices, the extend kernel reads them through kv_indices, and only the new tokens are
computed.  Slots referenced by the tree are owned by the tree: they return to the allocator only
through evict() (least recently used unlocked leaves first)."""
import heapq
import time
from typing import Dict, List, Optional, Tuple

import torch


class TreeNode:
    _counter = 0

    def __init__(self):
        self.children: Dict[int, "TreeNode"] = {}
        self.parent: Optional["TreeNode"] = None
        self.key: List[int] = []
        self.value: Optional[torch.Tensor] = None   # int64 slots, one per key token
        self.lock_ref = 0
        self.last_access_time = time.monotonic()
        TreeNode._counter += 1
        self.id = TreeNode._counter

    def __lt__(self, other: "TreeNode"):
        return self.last_access_time < other.last_access_time


def _match_len(a: List[int], b: List[int], page_size: int = 1) -> int:
    n = min(len(a), len(b))
    i = 0
    if page_size == 1:
        while i < n and a[i] == b[i]:
            i += 1
        return i
    while i + page_size <= n and a[i:i + page_size] == b[i:i + page_size]:
        i += page_size
    return i


class RadixCache:
    def __init__(self, req_to_token_pool, token_to_kv_pool_allocator, disable: bool = False, page_size: int = 1):
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool_allocator = token_to_kv_pool_allocator
        self.disable = disable
        self.page_size = page_size
        self.device = token_to_kv_pool_allocator.device
        self.reset()

    def reset(self):
        self.root_node = TreeNode()
        self.root_node.key = []
        self.root_node.value = torch.empty((0,), dtype=torch.int64, device=self.device)
        self.root_node.lock_ref = 1
        self.evictable_size_ = 0
        self.protected_size_ = 0

    def _child_key(self, key: List[int]):
        return key[0] if self.page_size == 1 else tuple(key[:self.page_size])

    def _aligned(self, n: int) -> int:
        return n // self.page_size * self.page_size

    # ---- lookup ----
    def match_prefix(self, key: List[int]) -> Tuple[torch.Tensor, TreeNode]:
        """Longest cached prefix of `key`: (its slots, the node it ends in).  A node that matches
        only partly is split so that the returned node ends exactly at the match (:134-178)."""
        if self.disable or len(key) == 0:
            return torch.empty((0,), dtype=torch.int64, device=self.device), self.root_node
        key = key[:self._aligned(len(key))]
        values, node = [], self.root_node
        now = time.monotonic()
        node.last_access_time = now
        while len(key) > 0 and self._child_key(key) in node.children:
            child = node.children[self._child_key(key)]
            child.last_access_time = now
            n = _match_len(child.key, key, self.page_size)
            if n < len(child.key):
                child = self._split_node(child, n)
                values.append(child.value)
                node = child
                break
            values.append(child.value)
            node = child
            key = key[n:]
        value = torch.cat(values) if values else torch.empty((0,), dtype=torch.int64, device=self.device)
        return value, node

    def _split_node(self, child: TreeNode, split_len: int) -> TreeNode:
        """child = [first split_len tokens] -> new parent node, rest stays in child (:380-398)."""
        new_node = TreeNode()
        new_node.children = {self._child_key(child.key[split_len:]): child}
        new_node.parent = child.parent
        new_node.lock_ref = child.lock_ref
        new_node.key = child.key[:split_len]
        new_node.value = child.value[:split_len]
        new_node.last_access_time = child.last_access_time
        child.parent = new_node
        child.key = child.key[split_len:]
        child.value = child.value[split_len:]
        new_node.parent.children[self._child_key(new_node.key)] = new_node
        return new_node

    # ---- insertion ----
    def insert(self, key: List[int], value: torch.Tensor) -> int:
        """Add key -> slots; returns how many leading tokens were already cached (their slots in
        `value` are duplicates the caller must free) (:180-186, _insert_helper :400-432)."""
        if self.disable:
            return 0
        node = self.root_node
        now = time.monotonic()
        node.last_access_time = now
        total_prefix = 0
        assert len(key) % self.page_size == 0, "insert whole pages"
        while len(key) > 0 and self._child_key(key) in node.children:
            child = node.children[self._child_key(key)]
            child.last_access_time = now
            n = _match_len(child.key, key, self.page_size)
            total_prefix += n
            key, value = key[n:], value[n:]
            if n < len(child.key):
                child = self._split_node(child, n)
            node = child
        if len(key) > 0:
            new_node = TreeNode()
            new_node.parent = node
            new_node.key = list(key)
            new_node.value = value.clone()
            node.children[self._child_key(key)] = new_node
            self.evictable_size_ += len(key)
        return total_prefix

    # ---- request life cycle ----
    def cache_finished_req(self, req, num_extra_slots: int = 0) -> None:
        """All but the last token of a finished request have KV: give them to the tree, free the
        duplicates and the request row, unlock its path (:188-222).  num_extra_slots: slots the
        request holds past those (a decode step launched before its end was known)."""
        token_ids = (req.origin_input_ids + req.output_ids)[:-1]
        pool = self.req_to_token_pool
        kv_indices = pool.req_to_token[req.req_pool_idx, : len(token_ids) + num_extra_slots].to(torch.int64)
        if self.disable:
            self.token_to_kv_pool_allocator.free(kv_indices)
            pool.free(req.req_pool_idx)
            return
        aligned = self._aligned(len(token_ids))
        if aligned < len(kv_indices):
            # the partly filled last page and the extra slots, in ONE call: a paged allocator takes
            # back whole pages and must not see the same page twice
            self.token_to_kv_pool_allocator.free(kv_indices[aligned:])
        new_prefix_len = self.insert(token_ids[:aligned], kv_indices[:aligned].clone())
        self.token_to_kv_pool_allocator.free(kv_indices[len(req.prefix_indices): new_prefix_len])
        pool.free(req.req_pool_idx)
        self.dec_lock_ref(req.last_node)

    def cache_unfinished_req(self, req) -> None:
        """A prompt that is only partly prefilled (chunked): cache what is computed, point the
        request at the tree's copy of its prefix, move its lock to the new node (:224-270)."""
        if self.disable:
            return
        token_ids = req.fill_ids
        pool = self.req_to_token_pool
        kv_indices = pool.req_to_token[req.req_pool_idx, : len(token_ids)].to(torch.int64)
        aligned = self._aligned(len(token_ids))
        new_prefix_len = self.insert(token_ids[:aligned], kv_indices[:aligned].clone())
        self.token_to_kv_pool_allocator.free(kv_indices[len(req.prefix_indices): new_prefix_len])
        new_indices, new_last_node = self.match_prefix(token_ids[:aligned])
        assert len(new_indices) == aligned
        if len(new_indices) > len(req.prefix_indices):
            pool.write((req.req_pool_idx, slice(len(req.prefix_indices), len(new_indices))),
                       new_indices[len(req.prefix_indices):].to(torch.int32))
        self.dec_lock_ref(req.last_node)
        self.inc_lock_ref(new_last_node)
        # the tree's slots for the whole pages, the request's own for the page still filling (:281-285)
        req.prefix_indices = torch.cat([new_indices, kv_indices[len(new_indices):]]) if aligned < len(token_ids) else new_indices
        req.last_node = new_last_node

    # ---- eviction / locking ----
    def evictable_size(self) -> int:
        return self.evictable_size_

    def protected_size(self) -> int:
        return self.protected_size_

    def evict(self, num_tokens: int) -> int:
        """Free at least num_tokens slots from unlocked leaves, least recently used first (:283-306)."""
        if self.disable:
            return 0
        leaves = [n for n in self._collect_leaves() if n.lock_ref == 0]
        heapq.heapify(leaves)
        freed = 0
        while freed < num_tokens and leaves:
            x = heapq.heappop(leaves)
            if x is self.root_node or x.lock_ref > 0:
                continue
            self.token_to_kv_pool_allocator.free(x.value)
            freed += len(x.value)
            self.evictable_size_ -= len(x.value)
            del x.parent.children[self._child_key(x.key)]
            if len(x.parent.children) == 0 and x.parent.lock_ref == 0 and x.parent is not self.root_node:
                heapq.heappush(leaves, x.parent)
        return freed

    def _collect_leaves(self) -> List[TreeNode]:
        out, stack = [], [self.root_node]
        while stack:
            n = stack.pop()
            if len(n.children) == 0:
                out.append(n)
            else:
                stack.extend(n.children.values())
        return out

    def inc_lock_ref(self, node: TreeNode) -> None:
        if self.disable:
            return
        while node is not self.root_node:
            if node.lock_ref == 0:
                self.evictable_size_ -= len(node.value)
                self.protected_size_ += len(node.value)
            node.lock_ref += 1
            node = node.parent

    def dec_lock_ref(self, node: TreeNode) -> None:
        if self.disable:
            return
        while node is not self.root_node:
            if node.lock_ref == 1:
                self.evictable_size_ += len(node.value)
                self.protected_size_ -= len(node.value)
            node.lock_ref -= 1
            node = node.parent

    def total_size(self) -> int:
        total, stack = 0, [self.root_node]
        while stack:
            n = stack.pop()
            total += len(n.value)
            stack.extend(n.children.values())
        return total
