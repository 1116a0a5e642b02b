"""torch.Tensor front-end of the C ABI (include/mi355_hotpath.h).

Each function checks device / dtype / contiguity (raising like the reference's
CHECK_INPUT / TORCH_CHECK do, sgl-kernel/csrc/torch_extension_rocm.cc), then passes raw
pointers and the current torch stream to libmi355_hotpath.so.  Nothing here computes: PyTorch
is used for memory and streams only.  Outputs are allocated by the caller or returned fresh,
mirroring sgl-kernel/python/sgl_kernel/{elementwise,sampling}.py.
"""
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import BF16, F16, F32, I32, I64, check

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
_IT = {torch.int32: I32, torch.int64: I64}


def _lib_():
    return _lib.load()


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}; expected float32/bfloat16/float16")


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mi355 hot-path ops need tensors on the GPU (no CPU fallback)")


KV_FP8 = 0x100   # MI355_KV_FP8: OR'ed into the dtype code when the KV cache holds e4m3fn bytes


KV_HEAD_MAJOR = 0x200   # MI355_KV_HEAD_MAJOR: the [slots, heads, d] cache tensors are views of [heads, slots, d] memory


KV_INTERLEAVED = 0x400  # MI355_KV_INTERLEAVED: ... of [heads, slots, 2, d] memory, keys in [..., 0, :], values in [..., 1, :]


def _kv_layout(buf: torch.Tensor) -> int:
    """0 for a contiguous [slots, heads, d] cache tensor, KV_HEAD_MAJOR for the permute(1, 0, 2) view of a
    contiguous [heads, slots, d] one, KV_HEAD_MAJOR | KV_INTERLEAVED for one half of a contiguous
    [heads, slots, 2, d] one seen the same way; anything else is refused."""
    if buf.is_contiguous():
        return 0
    assert buf.dim() == 3 and buf.stride(2) == 1, f"KV cache tensor with strides {buf.stride()}"
    s, _, d = buf.shape
    if buf.stride() == (d, s * d, 1):
        return KV_HEAD_MAJOR
    assert buf.stride() == (2 * d, 2 * s * d, 1), \
        f"KV cache tensor is neither [slots, heads, d] contiguous nor a head-major view: strides {buf.stride()}"
    return KV_HEAD_MAJOR | KV_INTERLEAVED


def _kvdt(t: torch.Tensor, cache: torch.Tensor, v_cache: Optional[torch.Tensor] = None) -> int:
    """dtype code of the 16-bit tensors, flagged when `cache` is an fp8 KV buffer and when it is head-major."""
    lay = _kv_layout(cache)
    if v_cache is not None:
        assert _kv_layout(v_cache) == lay and v_cache.shape[0] == cache.shape[0], "k and v cache layouts differ"
        if lay & KV_INTERLEAVED:
            assert (v_cache.shape == cache.shape
                    and v_cache.data_ptr() == cache.data_ptr() + cache.shape[2] * cache.element_size()), \
                "an interleaved cache holds a token's value row right behind its key row"
    return _dt(t) | (KV_FP8 if cache.dtype == torch.float8_e4m3fn else 0) | lay


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------- norms
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sgl_kernel.rmsnorm(input, weight, eps, out) (sgl-kernel/python/sgl_kernel/elementwise.py)."""
    _gpu(x, weight)
    assert x.dim() == 2 and x.stride(1) == 1 and weight.is_contiguous()
    assert weight.dtype == x.dtype and weight.numel() == x.shape[1]
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
    assert out.stride(1) == 1 and out.shape == x.shape and out.dtype == x.dtype
    if x.numel() == 0:
        return out
    check(_lib_().mi355_rmsnorm(out.data_ptr(), x.data_ptr(), weight.data_ptr(), x.shape[0],
                                x.shape[1], x.stride(0), out.stride(0), float(eps), _dt(x),
                                _stream(x)), "rmsnorm")
    return out


def fused_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                      eps: float) -> None:
    """In place on x and residual (sgl_kernel.fused_add_rmsnorm)."""
    _gpu(x, residual, weight)
    assert x.dim() == 2 and x.is_contiguous() and residual.is_contiguous()
    assert residual.shape == x.shape and residual.dtype == x.dtype == weight.dtype
    if x.numel() == 0:
        return
    check(_lib_().mi355_fused_add_rmsnorm(x.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                          x.shape[0], x.shape[1], float(eps), _dt(x), _stream(x)),
          "fused_add_rmsnorm")


def silu_and_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sgl_kernel.silu_and_mul(input, out): out = silu(x[..., :d]) * x[..., d:]."""
    _gpu(x)
    assert x.is_contiguous() and x.shape[-1] % 2 == 0
    d = x.shape[-1] // 2
    rows = x.numel() // (2 * d)
    if out is None:
        out = torch.empty(x.shape[:-1] + (d,), dtype=x.dtype, device=x.device)
    assert out.is_contiguous() and out.dtype == x.dtype
    if x.numel() == 0:
        return out
    check(_lib_().mi355_silu_and_mul(out.data_ptr(), x.data_ptr(), rows, d, _dt(x), _stream(x)),
          "silu_and_mul")
    return out


def softcap_inplace(logits: torch.Tensor, cap: float) -> torch.Tensor:
    """fused_softcap(full_logits, final_logit_softcapping) (logits_processor.py:607-618): in place,
    fp32, logits <- cap * tanh(logits / cap)."""
    _gpu(logits)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and cap > 0
    check(_lib_().mi355_softcap_inplace(logits.data_ptr(), logits.numel(), float(cap), _stream(logits)),
          "softcap_inplace")
    return logits


# ---------------------------------------------------------------------------------- rope / kv
def rope_inplace(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, head_size: int,
                 cos_sin_cache: torch.Tensor, is_neox: bool, v: Optional[torch.Tensor] = None,
                 k_buffer: Optional[torch.Tensor] = None, v_buffer: Optional[torch.Tensor] = None,
                 loc: Optional[torch.Tensor] = None) -> None:
    """apply_rope_with_cos_sin_cache_inplace(positions, query, key, head_size, cos_sin_cache,
    is_neox) (sgl-kernel/python/sgl_kernel/elementwise.py), optionally fused with the KV-cache
    write of MHATokenToKVPool.set_kv_buffer when k_buffer/v_buffer/loc/v are given.
    q: [T, Hq*D], k: [T, Hkv*D] (row-strided views allowed, last dim contiguous)."""
    _gpu(positions, q, k, cos_sin_cache)
    assert positions.dtype == torch.int64 and positions.is_contiguous()
    T = positions.numel()
    q2, k2 = q.view(T, -1), k.view(T, -1)
    assert q2.stride(1) == 1 and k2.stride(1) == 1
    hq, hkv = q2.shape[1] // head_size, k2.shape[1] // head_size
    rotary_dim = cos_sin_cache.shape[1]
    assert cos_sin_cache.is_contiguous()
    cache_dt = _dt(cos_sin_cache)
    fused = k_buffer is not None
    if fused:
        assert v is not None and v_buffer is not None and loc is not None
        v2 = v.view(T, -1)
        assert v2.stride(1) == 1
        assert k_buffer.dtype == q.dtype and v_buffer.dtype == q.dtype and v.dtype == q.dtype
        assert loc.is_contiguous() and loc.numel() == T
    check(_lib_().mi355_rope_inplace(
        q2.data_ptr(), k2.data_ptr(), _p(v) if fused else None, positions.data_ptr(),
        cos_sin_cache.data_ptr(), cache_dt, T, hq, hkv, head_size, rotary_dim, q2.stride(0),
        k2.stride(0), v.view(T, -1).stride(0) if fused else 0, 1 if is_neox else 0,
        _p(k_buffer) if fused else None, _p(v_buffer) if fused else None,
        _p(loc) if fused else None, _IT[loc.dtype] if fused else I64,
        k_buffer.shape[0] if fused else 0, _kvdt(q, k_buffer, v_buffer) if fused else _dt(q), _stream(q)), "rope_inplace")


def set_kv_buffer(k_buffer: torch.Tensor, v_buffer: torch.Tensor, loc: torch.Tensor,
                  cache_k: torch.Tensor, cache_v: torch.Tensor) -> None:
    """k_buffer[loc] = cache_k; v_buffer[loc] = cache_v (memory_pool.py:457-458)."""
    _gpu(k_buffer, v_buffer, loc, cache_k, cache_v)
    T = loc.numel()
    k2, v2 = cache_k.view(T, -1), cache_v.view(T, -1)
    assert k2.stride(1) == 1 and v2.stride(1) == 1 and loc.is_contiguous()
    assert cache_k.dtype == cache_v.dtype and k_buffer.dtype == v_buffer.dtype
    assert k_buffer.dtype in (cache_k.dtype, torch.float8_e4m3fn)
    k_row = k_buffer.numel() // k_buffer.shape[0]
    v_row = v_buffer.numel() // v_buffer.shape[0]
    assert k2.shape[1] == k_row and v2.shape[1] == v_row
    kvdt = _kvdt(cache_k, k_buffer, v_buffer)
    if kvdt & KV_HEAD_MAJOR:     # the scatter must know the heads: rows of the cache are not contiguous
        heads = k_buffer.shape[1]
        check(_lib_().mi355_set_kv_buffer_heads(k_buffer.data_ptr(), v_buffer.data_ptr(), k2.data_ptr(), v2.data_ptr(),
                                                loc.data_ptr(), _IT[loc.dtype], T, heads, k_buffer.shape[2],
                                                v_buffer.shape[2], k2.stride(0), v2.stride(0), k_buffer.shape[0],
                                                kvdt, _stream(cache_k)), "set_kv_buffer_heads")
        return
    check(_lib_().mi355_set_kv_buffer(k_buffer.data_ptr(), v_buffer.data_ptr(), k2.data_ptr(),
                                      v2.data_ptr(), loc.data_ptr(), _IT[loc.dtype], T, k_row,
                                      v_row, k2.stride(0), v2.stride(0), k_buffer.shape[0],
                                      kvdt, _stream(cache_k)), "set_kv_buffer")


def embedding(ids: torch.Tensor, table: torch.Tensor, vocab_start: int = 0,
              vocab_end: Optional[int] = None) -> torch.Tensor:
    _gpu(ids, table)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and table.is_contiguous()
    if vocab_end is None:
        vocab_end = vocab_start + table.shape[0]
    out = torch.empty((ids.numel(), table.shape[1]), dtype=table.dtype, device=table.device)
    if ids.numel() == 0:
        return out
    check(_lib_().mi355_embedding(out.data_ptr(), ids.data_ptr(), table.data_ptr(), ids.numel(),
                                  table.shape[1], vocab_start, vocab_end, _dt(table),
                                  _stream(table)), "embedding")
    return out


# ---------------------------------------------------------------------------------- index ops
def create_kv_indices(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                      page_kernel_lens: torch.Tensor, kv_indptr: torch.Tensor,
                      kv_start_idx: Optional[torch.Tensor], kv_indices: torch.Tensor) -> None:
    """create_flashinfer_kv_indices_triton[(bs,)](...) (layers/attention/utils.py:5-41)."""
    _gpu(req_to_token, req_pool_indices, page_kernel_lens, kv_indptr, kv_indices)
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    assert kv_indptr.dtype == torch.int32 and kv_indices.dtype == torch.int32
    assert kv_start_idx is None or kv_start_idx.dtype == torch.int32
    bs = req_pool_indices.numel()
    check(_lib_().mi355_create_kv_indices(
        req_to_token.data_ptr(), req_pool_indices.data_ptr(), _IT[req_pool_indices.dtype],
        page_kernel_lens.data_ptr(), _IT[page_kernel_lens.dtype], kv_indptr.data_ptr(),
        _p(kv_start_idx), kv_indices.data_ptr(), req_to_token.stride(0), bs,
        _stream(req_to_token)), "create_kv_indices")


def write_req_to_token(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                       pre_lens: torch.Tensor, seq_lens: torch.Tensor, extend_lens: torch.Tensor,
                       out_cache_loc: torch.Tensor) -> None:
    """write_req_to_token_pool_triton[(bs,)](...) (managers/schedule_batch.py:1900-1934)."""
    _gpu(req_to_token, req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc)
    for t in (req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc):
        assert t.dtype == torch.int64 and t.is_contiguous()
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    check(_lib_().mi355_write_req_to_token(
        req_to_token.data_ptr(), req_pool_indices.data_ptr(), pre_lens.data_ptr(),
        seq_lens.data_ptr(), extend_lens.data_ptr(), out_cache_loc.data_ptr(),
        req_to_token.stride(0), req_pool_indices.numel(), _stream(req_to_token)),
        "write_req_to_token")


def compute_position(extend_prefix_lens: torch.Tensor, extend_seq_lens: torch.Tensor,
                     extend_seq_lens_sum: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """compute_position_triton (model_executor/forward_batch_info.py:703-726)."""
    _gpu(extend_prefix_lens, extend_seq_lens)
    assert extend_prefix_lens.dtype == torch.int32 and extend_seq_lens.dtype == torch.int32
    bs = extend_seq_lens.numel()
    dev = extend_seq_lens.device
    positions = torch.empty(extend_seq_lens_sum, dtype=torch.int64, device=dev)
    extend_start_loc = torch.empty(bs, dtype=torch.int32, device=dev)
    check(_lib_().mi355_compute_position(extend_prefix_lens.data_ptr(), extend_seq_lens.data_ptr(),
                                         positions.data_ptr(), extend_start_loc.data_ptr(), bs,
                                         _stream(extend_seq_lens)), "compute_position")
    return positions, extend_start_loc


def prepare_decode(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                   seq_lens: torch.Tensor, out_cache_loc: torch.Tensor,
                   positions: torch.Tensor) -> None:
    """In place: seq_lens += 1; positions = clamp(seq_lens-1, 0);
    req_to_token[req_pool_indices, seq_lens-1] = out_cache_loc
    (managers/schedule_batch.py:1543-1573, forward_batch_info.py:776-778)."""
    _gpu(req_to_token, req_pool_indices, seq_lens, out_cache_loc, positions)
    for t in (req_pool_indices, seq_lens, out_cache_loc, positions):
        assert t.dtype == torch.int64 and t.is_contiguous()
    check(_lib_().mi355_prepare_decode(req_to_token.data_ptr(), req_pool_indices.data_ptr(),
                                       seq_lens.data_ptr(), out_cache_loc.data_ptr(),
                                       positions.data_ptr(), req_to_token.stride(0),
                                       seq_lens.numel(), _stream(seq_lens)), "prepare_decode")


def copy_decode_inputs(dsts, srcs, n: int) -> None:
    """dsts[i][:n] = srcs[i][:n] for up to five int64 device vectors in one launch (the static inputs
    of a captured decode step, cuda_graph_runner.py:652-676)."""
    import ctypes

    assert 1 <= len(dsts) == len(srcs) <= 5
    for d, s_ in zip(dsts, srcs):
        _gpu(d, s_)
        assert d.dtype == s_.dtype == torch.int64 and d.is_contiguous() and s_.is_contiguous()
        assert d.numel() >= n and s_.numel() >= n
    arr = ctypes.c_void_p * len(dsts)
    check(_lib_().mi355_copy_decode_inputs(arr(*[d.data_ptr() for d in dsts]), arr(*[s_.data_ptr() for s_ in srcs]),
                                           len(dsts), n, _stream(dsts[0])), "copy_decode_inputs")


def alloc_extend_pages(prefix_lens: torch.Tensor, seq_lens: torch.Tensor, last_loc: torch.Tensor,
                       free_pages: torch.Tensor, page_size: int, extend_num_tokens: int):
    """Page-aligned slots for an extend batch (allocator.py:276-360).  Returns (out_indices int64
    [extend_num_tokens], pages consumed from the front of free_pages).  The caller made sure
    free_pages holds enough pages."""
    _gpu(prefix_lens, seq_lens, last_loc, free_pages)
    for t in (prefix_lens, seq_lens, last_loc, free_pages):
        assert t.dtype == torch.int64 and t.is_contiguous()
    bs = seq_lens.numel()
    out = torch.empty(extend_num_tokens, dtype=torch.int64, device=seq_lens.device)
    totals = torch.zeros(2, dtype=torch.int64, device=seq_lens.device)
    check(_lib_().mi355_alloc_extend(prefix_lens.data_ptr(), seq_lens.data_ptr(), last_loc.data_ptr(),
                                     free_pages.data_ptr(), out.data_ptr(), totals.data_ptr(), bs, page_size,
                                     _stream(seq_lens)), "alloc_extend")
    pages, tokens = totals.tolist()
    assert tokens == extend_num_tokens, (tokens, extend_num_tokens)
    return out, pages


def alloc_decode_pages(seq_lens: torch.Tensor, last_loc: torch.Tensor, free_pages: torch.Tensor, page_size: int):
    """One slot per request for a decode step (allocator.py:364-401); seq_lens counts the new token.
    Returns (out_indices int64 [bs], pages consumed)."""
    _gpu(seq_lens, last_loc, free_pages)
    for t in (seq_lens, last_loc, free_pages):
        assert t.dtype == torch.int64 and t.is_contiguous()
    bs = seq_lens.numel()
    out = torch.empty(bs, dtype=torch.int64, device=seq_lens.device)
    totals = torch.zeros(2, dtype=torch.int64, device=seq_lens.device)
    check(_lib_().mi355_alloc_decode(seq_lens.data_ptr(), last_loc.data_ptr(), free_pages.data_ptr(), out.data_ptr(),
                                     totals.data_ptr(), bs, page_size, _stream(seq_lens)), "alloc_decode")
    return out, int(totals[0].item())


# ---------------------------------------------------------------------------------- attention
def _tickets(t, need: int):
    """Device pointer of the split-merge arrival counters (None: the splits are merged by a second launch)."""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and t.numel() >= need):
        raise _lib.Mi355Error(f"split tickets must be >= {need} contiguous int32 counters on the GPU")
    return t.data_ptr()


def _chunk(t):
    """Device pointer of the batch's keys-per-split value (None: every sequence is cut into num_kv_splits)."""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.int32 and t.numel() >= 1):
        raise _lib.Mi355Error("kv_chunk_keys must be an int32 tensor on the GPU")
    return t.data_ptr()


def _split_scratch(attn_logits, attn_lse, rows: int, dv: int, splits: int) -> None:
    """The kernels write rows * splits partial outputs of dv floats and as many log-sum-exps: refuse a
    scratch pair that cannot hold them (the C ABI sees raw pointers and cannot).  Raises Mi355Error
    before anything is launched; not an assert, so that `python -O` keeps the refusal."""
    if splits <= 1:
        return
    if attn_logits is None or attn_lse is None:
        raise _lib.Mi355Error(f"num_kv_splits = {splits} needs attn_logits / attn_lse scratch")
    if not (attn_logits.is_cuda and attn_lse.is_cuda and attn_logits.dtype == attn_lse.dtype == torch.float32
            and attn_logits.is_contiguous() and attn_lse.is_contiguous()):
        raise _lib.Mi355Error("split scratch must be contiguous float32 tensors on the GPU")
    if attn_logits.numel() < rows * splits * dv or attn_lse.numel() < rows * splits:
        raise _lib.Mi355Error(
            f"split scratch too small for {rows} rows x {splits} splits x {dv}: attn_logits holds "
            f"{attn_logits.numel()} floats (needs {rows * splits * dv}), attn_lse {attn_lse.numel()} (needs {rows * splits})")


def decode_num_splits(bs: int, hkv: int, max_seq_len: int, max_splits: int) -> int:
    return _lib_().mi355_decode_num_splits(bs, hkv, max_seq_len, max_splits)


def decode_attention_fwd(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse,
                         num_kv_splits: int, max_kv_splits: int, sm_scale: float,
                         logit_cap: float = 0.0, tickets=None, kv_chunk_keys=None) -> None:
    """decode_attention_fwd(...) of the Triton backend (triton_ops/decode_attention.py:677-728),
    kv_indptr/kv_indices addressing.  num_kv_splits is one int for the whole batch.
    tickets: see decode_attention_paged."""
    _gpu(q, k_buffer, v_buffer, o, kv_indptr, kv_indices)
    B, Hq, D = q.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    assert q.stride(2) == 1 and q.stride(1) == D and o.is_contiguous()
    assert kv_indptr.dtype == torch.int32 and kv_indices.dtype == torch.int32
    assert num_kv_splits <= max_kv_splits
    _split_scratch(attn_logits, attn_lse, B * Hq, Dv, num_kv_splits)
    check(_lib_().mi355_decode_attention(
        q.data_ptr(), k_buffer.data_ptr(), v_buffer.data_ptr(), o.data_ptr(),
        kv_indices.data_ptr(), 1, None, None, kv_indptr.data_ptr(), 0, _p(attn_logits),
        _p(attn_lse), _tickets(tickets, B * Hq), _chunk(kv_chunk_keys), num_kv_splits, B, Hq, Hkv, D, Dv, q.stride(0), k_buffer.shape[0], float(sm_scale),
        float(logit_cap), _kvdt(q, k_buffer, v_buffer), _stream(q)), "decode_attention")


def decode_attention_paged(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens,
                           attn_logits, attn_lse, num_kv_splits: int, sm_scale: float,
                           logit_cap: float = 0.0, tickets=None, kv_chunk_keys=None) -> None:
    """decode_attention_cpu-style addressing (sgl-kernel/csrc/cpu/decode.cpp:1375): the kernel
    reads req_to_token[req_pool_indices[b], :seq_lens[b]] directly, no kv_indices gather.
    tickets (optional): >= B * Hq int32 arrival counters, zero between calls (the kernel leaves them
    zero); with them the KV splits are merged inside the attention launch instead of by a second one."""
    _gpu(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens)
    B, Hq, D = q.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    assert q.stride(2) == 1 and q.stride(1) == D and o.is_contiguous()
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    assert req_pool_indices.dtype == torch.int64 and seq_lens.dtype == torch.int64
    _split_scratch(attn_logits, attn_lse, B * Hq, Dv, num_kv_splits)
    check(_lib_().mi355_decode_attention(
        q.data_ptr(), k_buffer.data_ptr(), v_buffer.data_ptr(), o.data_ptr(),
        req_to_token.data_ptr(), 0, req_pool_indices.data_ptr(), seq_lens.data_ptr(), None,
        req_to_token.stride(0), _p(attn_logits), _p(attn_lse), _tickets(tickets, B * Hq), _chunk(kv_chunk_keys), num_kv_splits, B, Hq, Hkv, D, Dv,
        q.stride(0), k_buffer.shape[0], float(sm_scale), float(logit_cap), _kvdt(q, k_buffer, v_buffer), _stream(q)),
        "decode_attention")


def decode_attention_paged_static(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens,
                                  num_kv_splits: int, sm_scale: float, logit_cap: float = 0.0) -> None:
    """decode_attention_paged without caller scratch: the split partials live in the library's own
    fixed device array (mi355_decode_attention_static), so the call allocates nothing.  Calls that
    may overlap on one GPU race on that array."""
    _gpu(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens)
    B, Hq, D = q.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    assert q.stride(2) == 1 and q.stride(1) == D and o.is_contiguous()
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    assert req_pool_indices.dtype == torch.int64 and seq_lens.dtype == torch.int64
    check(_lib_().mi355_decode_attention_static(
        q.data_ptr(), k_buffer.data_ptr(), v_buffer.data_ptr(), o.data_ptr(),
        req_to_token.data_ptr(), 0, req_pool_indices.data_ptr(), seq_lens.data_ptr(), None,
        req_to_token.stride(0), num_kv_splits, B, Hq, Hkv, D, Dv,
        q.stride(0), k_buffer.shape[0], float(sm_scale), float(logit_cap), _kvdt(q, k_buffer, v_buffer), _stream(q)),
        "decode_attention_static")


def decode_attention_fused_qkv(qkv_part, positions, out_cache_loc, cos_sin_cache, k_buffer, v_buffer, o, req_to_token,
                               req_pool_indices, seq_lens, attn_logits, attn_lse, num_kv_splits: int,
                               sm_scale: float, part_x_scale=None, part_w_scale=None, tickets=None, kv_chunk_keys=None) -> None:
    """Decode attention straight from the qkv GEMM's split-K partials [splits, B, (Hq + 2 Hkv) * D]:
    sums them, applies neox RoPE to q and the new k, writes the new k / v rows into cache slot
    out_cache_loc[b] (= req_to_token[req_pool_indices[b], seq_lens[b] - 1] for real rows) and attends.  o: [B, Hq, D]."""
    _gpu(qkv_part, positions, cos_sin_cache, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens)
    B, Hq, D = o.shape
    Hkv = k_buffer.shape[1]
    assert qkv_part.dim() == 3 and qkv_part.is_contiguous() and qkv_part.dtype == torch.float32
    assert qkv_part.shape[1] == B and qkv_part.shape[2] == (Hq + 2 * Hkv) * D
    assert o.is_contiguous()
    assert k_buffer.shape[1:] == (Hkv, D) and v_buffer.shape == k_buffer.shape
    assert k_buffer.dtype == v_buffer.dtype and k_buffer.dtype in (o.dtype, torch.float8_e4m3fn)
    assert positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == B
    assert out_cache_loc.dtype == torch.int64 and out_cache_loc.is_contiguous() and out_cache_loc.numel() == B
    assert cos_sin_cache.is_contiguous() and cos_sin_cache.shape[1] == D
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    assert req_pool_indices.dtype == torch.int64 and seq_lens.dtype == torch.int64
    if part_x_scale is not None:   # partials of an fp8 GEMM: its per-token / per-channel scales
        assert part_x_scale.dtype == part_w_scale.dtype == torch.float32 and part_x_scale.numel() == B
        assert part_w_scale.numel() == qkv_part.shape[2] and part_x_scale.is_contiguous() and part_w_scale.is_contiguous()
    _split_scratch(attn_logits, attn_lse, B * Hq, D, num_kv_splits)
    check(_lib_().mi355_decode_attention_fused_qkv(
        qkv_part.data_ptr(), qkv_part.shape[0], _p(part_x_scale), _p(part_w_scale), positions.data_ptr(),
        out_cache_loc.data_ptr(), cos_sin_cache.data_ptr(),
        _dt(cos_sin_cache), k_buffer.data_ptr(), v_buffer.data_ptr(), o.data_ptr(), req_to_token.data_ptr(), 0,
        req_pool_indices.data_ptr(), seq_lens.data_ptr(), None, req_to_token.stride(0), _p(attn_logits),
        _p(attn_lse), _tickets(tickets, B * Hq), _chunk(kv_chunk_keys), num_kv_splits, B, Hq, Hkv, D, k_buffer.shape[0], float(sm_scale), _kvdt(o, k_buffer, v_buffer), _stream(o)),
        "decode_attention_fused_qkv")


def decode_attention_mla(q, kv_buffer, o, req_to_token, req_pool_indices, seq_lens, attn_logits,
                         attn_lse, num_kv_splits: int, kv_lora_rank: int, sm_scale: float,
                         logit_cap: float = 0.0, tickets=None, kv_chunk_keys=None) -> None:
    """MQA over the compressed MLA cache (deepseek_v2.py:866-875): kv_buffer [slots, 1, rank + rope],
    q [B, H, rank + rope], o [B, H, rank]; the value is kv_buffer[..., :rank].  kv_buffer may hold
    e4m3fn bytes (kv_cache_dtype fp8_e4m3): the rows are widened exactly on use."""
    _gpu(q, kv_buffer, o, req_to_token, req_pool_indices, seq_lens)
    B, Hq, D = q.shape
    assert kv_buffer.is_contiguous() and kv_buffer.shape[1] == 1 and kv_buffer.shape[2] == D
    assert q.stride(2) == 1 and q.stride(1) == D and o.is_contiguous() and o.shape == (B, Hq, kv_lora_rank)
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    assert req_pool_indices.dtype == torch.int64 and seq_lens.dtype == torch.int64
    _split_scratch(attn_logits, attn_lse, B * Hq, kv_lora_rank, num_kv_splits)
    check(_lib_().mi355_decode_attention_mla(
        q.data_ptr(), kv_buffer.data_ptr(), o.data_ptr(), req_to_token.data_ptr(), 0,
        req_pool_indices.data_ptr(), seq_lens.data_ptr(), None, req_to_token.stride(0),
        _p(attn_logits), _p(attn_lse), _tickets(tickets, B * Hq), _chunk(kv_chunk_keys), num_kv_splits, B, Hq, kv_lora_rank, D - kv_lora_rank,
        q.stride(0), kv_buffer.shape[0], float(sm_scale), float(logit_cap), _kvdt(q, kv_buffer), _stream(q)),
        "decode_attention_mla")


def decode_attention_mla_fused_rope_supported(kv_lora_rank: int, rope_dim: int, dtype: torch.dtype, cache_dtype: torch.dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16) and bool(_lib_().mi355_decode_attention_mla_fused_rope_supported(
        kv_lora_rank, rope_dim, _DT[dtype], 1 if cache_dtype == torch.float8_e4m3fn else 0))


def decode_attention_mla_fused_rope(q, k_nope, k_pe, positions, out_cache_loc, cos_sin_cache, is_neox: bool, kv_buffer, o,
                                    req_to_token, req_pool_indices, seq_lens, attn_logits, attn_lse, num_kv_splits: int,
                                    kv_lora_rank: int, sm_scale: float, logit_cap: float = 0.0, tickets=None,
                                    kv_chunk_keys=None) -> None:
    """decode_attention_mla with RoPE and the cache write inside the launch (the reference's ROCm kernel
    decode_attention_fwd_grouped_rope, rocm_mla_decode_rope.py:402-439): q [B, H, rank + rope] with its rope part
    NOT rotated, k_nope [B, rank] (row-strided view allowed), k_pe [B, rope] not rotated (row-strided view allowed);
    rotates both at positions, writes the row [k_nope | rope(k_pe)] to kv_buffer[out_cache_loc] (quantised as
    set_kv_buffer does when the cache is e4m3fn) and attends."""
    _gpu(q, k_nope, k_pe, positions, out_cache_loc, cos_sin_cache, kv_buffer, o, req_to_token, req_pool_indices, seq_lens)
    B, Hq, D = q.shape
    rope = D - kv_lora_rank
    assert kv_buffer.is_contiguous() and kv_buffer.shape[1] == 1 and kv_buffer.shape[2] == D
    assert kv_buffer.dtype in (q.dtype, torch.float8_e4m3fn)
    assert q.stride(2) == 1 and q.stride(1) == D and o.is_contiguous() and o.shape == (B, Hq, kv_lora_rank)
    k_nope2, k_pe2 = k_nope.view(B, -1), k_pe.view(B, -1)
    assert k_nope2.shape[1] == kv_lora_rank and k_pe2.shape[1] == rope and k_nope2.stride(1) == 1 and k_pe2.stride(1) == 1
    assert k_nope2.dtype == q.dtype and k_pe2.dtype == q.dtype
    assert positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == B
    assert out_cache_loc.dtype == torch.int64 and out_cache_loc.is_contiguous() and out_cache_loc.numel() == B
    assert cos_sin_cache.is_contiguous() and cos_sin_cache.shape[1] == rope
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    assert req_pool_indices.dtype == torch.int64 and seq_lens.dtype == torch.int64
    _split_scratch(attn_logits, attn_lse, B * Hq, kv_lora_rank, num_kv_splits)
    check(_lib_().mi355_decode_attention_mla_fused_rope(
        q.data_ptr(), k_nope2.data_ptr(), k_nope2.stride(0), k_pe2.data_ptr(), k_pe2.stride(0), positions.data_ptr(),
        out_cache_loc.data_ptr(), cos_sin_cache.data_ptr(), _dt(cos_sin_cache), 1 if is_neox else 0, kv_buffer.data_ptr(),
        o.data_ptr(), req_to_token.data_ptr(), 0, req_pool_indices.data_ptr(), seq_lens.data_ptr(), None,
        req_to_token.stride(0), _p(attn_logits), _p(attn_lse), _tickets(tickets, B * Hq), _chunk(kv_chunk_keys),
        num_kv_splits, B, Hq, kv_lora_rank, rope, q.stride(0), kv_buffer.shape[0], float(sm_scale), float(logit_cap),
        _kvdt(q, kv_buffer), _stream(q)), "decode_attention_mla_fused_rope")


def mla_latent_norms_supported(q_rank: int, kv_rank: int, rope_dim: int, dtype: torch.dtype) -> bool:
    """Whether mla_latent_norms serves the shape (16-bit; the widths multiples of 8, at most 2048 per part)."""
    return (dtype in (torch.bfloat16, torch.float16) and q_rank % 8 == 0 and kv_rank % 8 == 0 and rope_dim % 8 == 0
            and 0 <= q_rank <= 2048 and 0 < kv_rank and 0 < rope_dim and kv_rank + rope_dim <= 2048)


def mla_latent_norms(q_weight: Optional[torch.Tensor], kv_weight: torch.Tensor, q_rank: int, kv_rank: int, rope_dim: int,
                     eps: float, x: Optional[torch.Tensor] = None, part: Optional[torch.Tensor] = None,
                     part_x_scale: Optional[torch.Tensor] = None, part_w_scale: Optional[torch.Tensor] = None,
                     quant_q_fp8: bool = False):
    """q_a_layernorm and kv_a_layernorm over the two parts of fused_qkv_a_proj_with_mqa's output row
    [q_a | c_kv | k_pe] in one launch (deepseek_v2.py:1193-1232): from x [M, width] (row-strided view allowed) or
    from that GEMM's split-K partials [splits, M, width] (with the fp8 GEMM's scales when it was one).  Returns
    (q_a normed [M, q_rank] or None when q_rank == 0, c_kv normed [M, kv_rank], k_pe [M, rope_dim]), all dense;
    with quant_q_fp8 additionally (e4m3fn [M, q_rank], scale [M, 1]) = per_token_quant_fp8 of the normed q_a."""
    src = part if part is not None else x
    _gpu(src, q_weight, kv_weight)
    dt = kv_weight.dtype
    width = q_rank + kv_rank + rope_dim
    if part is not None:
        assert part.dim() == 3 and part.is_contiguous() and part.dtype == torch.float32 and part.shape[2] == width
        rows = part.shape[1]
    else:
        assert x.dim() == 2 and x.shape[1] == width and x.stride(1) == 1 and x.dtype == dt
        rows = x.shape[0]
    assert kv_weight.is_contiguous() and kv_weight.numel() == kv_rank
    assert (q_weight is None) == (q_rank == 0)
    q_out = None
    if q_rank:
        assert q_weight.is_contiguous() and q_weight.numel() == q_rank and q_weight.dtype == dt
        q_out = torch.empty((rows, q_rank), dtype=dt, device=kv_weight.device)
    kv_out = torch.empty((rows, kv_rank), dtype=dt, device=kv_weight.device)
    pe_out = torch.empty((rows, rope_dim), dtype=dt, device=kv_weight.device)
    q8 = s8 = None
    if quant_q_fp8:
        assert q_rank > 0
        q8 = torch.empty((rows, q_rank), dtype=torch.float8_e4m3fn, device=kv_weight.device)
        s8 = torch.empty((rows, 1), dtype=torch.float32, device=kv_weight.device)
    if rows:
        check(_lib_().mi355_mla_latent_norms(
            _p(q_out), kv_out.data_ptr(), pe_out.data_ptr(), _p(x), x.stride(0) if x is not None else 0, _p(part),
            part.shape[0] if part is not None else 0, _p(part_x_scale), _p(part_w_scale), _p(q_weight),
            kv_weight.data_ptr(), _p(q8), _p(s8), rows, q_rank, kv_rank, rope_dim, float(eps), _dt(kv_weight),
            _stream(kv_weight)), "mla_latent_norms")
    if quant_q_fp8:
        return q_out, kv_out, pe_out, q8, s8
    return q_out, kv_out, pe_out


def mla_absorb_supported(m: int, n: int, k: int, tail: int, dtype: torch.dtype) -> bool:
    """Whether mla_absorb serves the shape (decode-sized batch, 16-bit, MFMA-tileable)."""
    return (dtype in (torch.bfloat16, torch.float16) and 0 <= m <= 64 and n >= 16 and n % 16 == 0 and k >= 32
            and k % 32 == 0 and tail % 8 == 0)


def mla_absorb(a: torch.Tensor, w_t: torch.Tensor, tail: int = 0) -> torch.Tensor:
    """The absorbed products of MLA decode as one launch each (deepseek_v2.py:1234-1262, 1296-1333):
    a [M, H, k + tail] (row- and head-strided view allowed, features contiguous), w_t [H, n, k] (each head's
    matrix with the reduction dimension contiguous) -> out [M, H, n + tail] with out[..., :n] = a[..., :k] @ w_t[h].T
    per head and out[..., n:] = a[..., k:].  M <= 64."""
    _gpu(a, w_t)
    m, heads, feat = a.shape
    hw, n, k = w_t.shape
    assert hw == heads and feat == k + tail and a.stride(2) == 1 and w_t.is_contiguous() and a.dtype == w_t.dtype
    out = torch.empty((m, heads, n + tail), dtype=a.dtype, device=a.device)
    if m:
        check(_lib_().mi355_mla_absorb(out.data_ptr(), a.data_ptr(), w_t.data_ptr(), m, heads, n, k, tail, a.stride(0),
                                       a.stride(1), out.stride(0), out.stride(1), _dt(a), _stream(a)), "mla_absorb")
    return out


def mla_prefill_assemble_supported(nope: int, rope: int, vd: int, rank: int, dtype: torch.dtype) -> bool:
    """Whether mla_prefill_assemble serves the shape (16-bit; widths multiples of 8, rope <= 256)."""
    return (dtype in (torch.bfloat16, torch.float16) and nope % 8 == 0 and rope % 8 == 0 and 0 < rope <= 256
            and vd % 8 == 0 and rank % 8 == 0)


def mla_prefill_assemble(q, kv, k_pe, c_kv, positions, cos_sin_cache, is_neox: bool, kv_cache, out_cache_loc,
                         nope: int, rope: int, vd: int):
    """The glue of MLA's multi-head prefill (deepseek_v2.py:1131-1179) in one launch: q [T, H, nope + rope] gets its
    rope part rotated in place; returns k [T, H, nope + rope] = [kv nope part | rope(k_pe)] and v [T, H, vd] from
    kv [T, H, nope + vd]; writes the latent row [c_kv | rope(k_pe)] to kv_cache[out_cache_loc] (quantised when the
    cache is e4m3fn).  k_pe [T, rope] and c_kv [T, rank] may be row-strided views."""
    _gpu(q, kv, k_pe, c_kv, positions, cos_sin_cache, kv_cache, out_cache_loc)
    T, H, D = q.shape
    rank = c_kv.shape[1]
    assert D == nope + rope and q.stride(2) == 1 and q.stride(1) == D
    assert kv.is_contiguous() and kv.shape == (T, H, nope + vd) and kv.dtype == q.dtype
    assert k_pe.shape == (T, rope) and k_pe.stride(1) == 1 and c_kv.shape[0] == T and c_kv.stride(1) == 1
    assert k_pe.dtype == q.dtype and c_kv.dtype == q.dtype
    assert positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == T
    assert out_cache_loc.dtype == torch.int64 and out_cache_loc.is_contiguous() and out_cache_loc.numel() == T
    assert cos_sin_cache.is_contiguous() and cos_sin_cache.shape[1] == rope
    assert kv_cache.is_contiguous() and kv_cache.shape[1] == 1 and kv_cache.shape[2] == rank + rope
    assert kv_cache.dtype in (q.dtype, torch.float8_e4m3fn)
    k = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    v = torch.empty((T, H, vd), dtype=q.dtype, device=q.device)
    if T:
        check(_lib_().mi355_mla_prefill_assemble(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), kv.data_ptr(), k_pe.data_ptr(), k_pe.stride(0), c_kv.data_ptr(),
            c_kv.stride(0), positions.data_ptr(), cos_sin_cache.data_ptr(), _dt(cos_sin_cache), 1 if is_neox else 0,
            kv_cache.data_ptr(), out_cache_loc.data_ptr(), T, H, nope, rope, vd, rank, q.stride(0), kv_cache.shape[0],
            _kvdt(q, kv_cache), _stream(q)), "mla_prefill_assemble")
    return k, v


def extend_attention_mla_supported(hq: int, kv_lora_rank: int, rope_dim: int, dtype: torch.dtype) -> bool:
    """Whether extend_attention_mla serves the shape (latent row of 512 + 64 features, 16-bit)."""
    return dtype in (torch.bfloat16, torch.float16) and bool(
        _lib_().mi355_extend_attention_mla_supported(hq, kv_lora_rank, rope_dim, _DT[dtype]))


def extend_attention_mla(q, kv_buffer, o, req_to_token, req_pool_indices, seq_lens, qo_indptr,
                         max_extend_len: int, kv_lora_rank: int, sm_scale: float, logit_cap: float = 0.0) -> None:
    """attn_mqa (deepseek_v2.py:866-875) of an extend batch in the absorbed form: request b's new tokens
    (rows qo_indptr[b] .. qo_indptr[b + 1] of q [T, H, rank + rope] and o [T, H, rank]) attend to the
    request's first seq_lens[b] cache rows causally; the new tokens' latent rows are in kv_buffer
    [slots, 1, rank + rope] already.  Four consecutive tokens share one pass over the rows."""
    _gpu(q, kv_buffer, o, req_to_token, req_pool_indices, seq_lens, qo_indptr)
    T, Hq, D = q.shape
    B = req_pool_indices.numel()
    assert kv_buffer.is_contiguous() and kv_buffer.shape[1] == 1 and kv_buffer.shape[2] == D
    assert q.stride(2) == 1 and q.stride(1) == D and o.is_contiguous() and o.shape == (T, Hq, kv_lora_rank)
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    assert req_pool_indices.dtype == torch.int64 and seq_lens.dtype == torch.int64 and seq_lens.numel() == B
    assert qo_indptr.dtype == torch.int32 and qo_indptr.is_contiguous() and qo_indptr.numel() == B + 1
    check(_lib_().mi355_extend_attention_mla(
        q.data_ptr(), kv_buffer.data_ptr(), o.data_ptr(), req_to_token.data_ptr(), req_pool_indices.data_ptr(),
        seq_lens.data_ptr(), qo_indptr.data_ptr(), req_to_token.stride(0), B, int(max_extend_len), Hq, kv_lora_rank,
        D - kv_lora_rank, q.stride(0), kv_buffer.shape[0], float(sm_scale), float(logit_cap), _dt(q), _stream(q)),
        "extend_attention_mla")


def extend_attention_fwd(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr,
                         kv_indptr, kv_indices, custom_mask, is_causal, mask_indptr,
                         max_len_extend, sm_scale=None, logit_cap=0.0,
                         skip_prefix_custom_mask=True, sliding_window_size=-1) -> None:
    """extend_attention_fwd(...) (triton_ops/extend_attention.py:306-438), same argument order."""
    _gpu(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices)
    T, Hq, D = q_extend.shape
    Hkv, Dv = v_extend.shape[1], v_extend.shape[2]
    assert q_extend.stride(2) == 1 and q_extend.stride(1) == D
    assert k_extend.stride(2) == 1 and k_extend.stride(1) == D
    assert v_extend.stride(2) == 1 and v_extend.stride(1) == Dv
    assert o_extend.is_contiguous()
    assert qo_indptr.dtype == torch.int32 and kv_indptr.dtype == torch.int32 and kv_indices.dtype == torch.int32
    if sm_scale is None:
        sm_scale = 1.0 / (D ** 0.5)
    if custom_mask is not None:
        assert mask_indptr is not None and mask_indptr.dtype == torch.int64
        if custom_mask.dtype == torch.bool:
            custom_mask = custom_mask.view(torch.uint8)
        assert custom_mask.dtype == torch.uint8
    bs = qo_indptr.numel() - 1
    check(_lib_().mi355_extend_attention(
        q_extend.data_ptr(), k_extend.data_ptr(), v_extend.data_ptr(), o_extend.data_ptr(),
        k_buffer.data_ptr(), v_buffer.data_ptr(), qo_indptr.data_ptr(), kv_indptr.data_ptr(),
        kv_indices.data_ptr(), _p(custom_mask), _p(mask_indptr) if custom_mask is not None else None,
        1 if is_causal else 0, 1 if skip_prefix_custom_mask else 0, bs, int(max_len_extend), Hq,
        Hkv, D, Dv, q_extend.stride(0), k_extend.stride(0), v_extend.stride(0),
        k_buffer.shape[0], float(sm_scale), float(logit_cap), int(sliding_window_size),
        _kvdt(q_extend, k_buffer, v_buffer), _stream(q_extend)), "extend_attention")


# ---------------------------------------------------------------------------------- linear
_WS = {}
_WS_RETIRED = []


def _workspace(device, nbytes: int, tag: str = "linear") -> torch.Tensor:
    """Per-device, per-kernel-family grow-only scratch for split-K partials (allocated outside
    graph capture by the first eager call of each shape; ModelRunner warms every shape up before
    capturing).  Zero-filled at allocation: the 16-bit GEMM keeps its arrival counters at the head
    of the buffer and hands them back zeroed after every launch.  Launches that share a buffer
    must be ordered on one stream."""
    ws = _WS.get((device, tag))
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            # a captured HIP graph may still hold the old pointer: outgrown buffers stay alive
            _WS_RETIRED.append(ws)
        ws = torch.zeros(max(nbytes, 8 << 20), dtype=torch.uint8, device=device)
        _WS[(device, tag)] = ws
    return ws


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """F.linear(x, weight, bias) (layers/quantization/unquant.py:110-122) through the C ABI:
    weight-streaming MFMA kernel for M <= 64 (bf16/f16), exact tiled kernel otherwise."""
    _gpu(x, weight, bias)
    assert x.dim() == 2 and x.stride(1) == 1 and weight.is_contiguous() and weight.dtype == x.dtype
    M, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K
    out_dtype = out_dtype or x.dtype
    y = torch.empty((M, N), dtype=out_dtype, device=x.device)
    if M == 0:
        return y
    lib = _lib_()
    need = lib.mi355_linear_workspace_bytes(M, N, K)
    ws = _workspace(x.device, need) if need > 0 else None
    check(lib.mi355_linear(y.data_ptr(), x.data_ptr(), weight.data_ptr(), _p(bias), M, N, K,
                           x.stride(0), _p(ws), ws.numel() if ws is not None else 0, _dt(x),
                           _DT[out_dtype], _stream(x)), "linear")
    return y


def linear_tiled_supported(m: int, n: int, k: int, dtype: torch.dtype, silu_mul: bool = False) -> bool:
    """Whether linear_tiled serves the shape (16-bit, K % 64 == 0, K >= 128, N % 16 == 0)."""
    return dtype in (torch.bfloat16, torch.float16) and bool(
        _lib_().mi355_linear_tiled_supported(m, n, k, _DT[dtype], 1 if silu_mul else 0))


def linear_tiled(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                 silu_mul: bool = False) -> torch.Tensor:
    """Prefill-sized F.linear(x, weight, bias) (layers/quantization/unquant.py:110-122) on the
    256 x 256-tile matrix-core kernel; silu_mul=True applies SiluAndMul (layers/activation.py:55-65)
    to the [gate; up] output in the epilogue and returns [M, N / 2]."""
    _gpu(x, weight, bias)
    assert x.dim() == 2 and x.stride(1) == 1 and weight.is_contiguous() and weight.dtype == x.dtype
    M, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K
    y = torch.empty((M, N // 2 if silu_mul else N), dtype=x.dtype, device=x.device)
    if M == 0:
        return y
    check(_lib_().mi355_linear_tiled(y.data_ptr(), x.data_ptr(), weight.data_ptr(), _p(bias), M, N, K,
                                     x.stride(0), _dt(x), 1 if silu_mul else 0, _stream(x)), "linear_tiled")
    return y


def qkv_proj_rope_tiled_supported(m: int, k: int, hq: int, hkv: int, head_dim: int, rotary_dim: int,
                                  is_neox: bool, dtype: torch.dtype, fp8: bool = False) -> bool:
    """Whether qkv_proj_rope_tiled serves the layer (head_dim = rotary_dim = 128, neox, 16-bit)."""
    return dtype in (torch.bfloat16, torch.float16) and bool(_lib_().mi355_qkv_proj_rope_tiled_supported(
        m, k, hq, hkv, head_dim, rotary_dim, 1 if is_neox else 0, _DT[dtype], 1 if fp8 else 0))


def qkv_proj_rope_tiled(x: torch.Tensor, weight: torch.Tensor, hq: int, hkv: int, head_dim: int,
                        positions: torch.Tensor, cos_sin_cache: torch.Tensor, is_neox: bool,
                        k_buffer: torch.Tensor, v_buffer: torch.Tensor, loc: torch.Tensor,
                        x_scale: Optional[torch.Tensor] = None, w_scale: Optional[torch.Tensor] = None,
                        out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """qkv_proj -> rotary_emb -> set_kv_buffer of a prefill batch (models/llama.py:187-199) as one
    kernel: the 256 x 256-tile projection with neox RoPE and the cache write in its epilogue.
    Returns qkv [M, (hq + 2 hkv) * head_dim] holding rotated q, rotated k and v.  With x_scale / w_scale
    x and weight are e4m3fn (per-token / per-channel scales) and out_dtype names the result dtype."""
    _gpu(x, weight, positions, cos_sin_cache, k_buffer, v_buffer, loc)
    fp8 = x_scale is not None
    dtype = out_dtype if fp8 else x.dtype
    assert x.dim() == 2 and x.stride(1) == 1 and weight.is_contiguous() and weight.dtype == x.dtype
    M, K = x.shape
    N = (hq + 2 * hkv) * head_dim
    assert weight.shape == (N, K)
    assert positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == M
    assert cos_sin_cache.is_contiguous() and cos_sin_cache.dim() == 2
    assert k_buffer.dtype == dtype and v_buffer.dtype == dtype
    assert k_buffer[0].numel() == hkv * head_dim and v_buffer[0].numel() == hkv * head_dim
    assert _kv_layout(k_buffer) == _kv_layout(v_buffer) and k_buffer.shape == v_buffer.shape
    assert loc.is_contiguous() and loc.numel() == M
    if fp8:
        assert x.dtype == torch.float8_e4m3fn and x.is_contiguous()
        assert x_scale.dtype == torch.float32 and x_scale.is_contiguous() and x_scale.numel() == M
        assert w_scale.dtype == torch.float32 and w_scale.is_contiguous() and w_scale.numel() == N
    qkv = torch.empty((M, N), dtype=dtype, device=x.device)
    if M == 0:
        return qkv
    check(_lib_().mi355_qkv_proj_rope_tiled(
        qkv.data_ptr(), x.data_ptr(), weight.data_ptr(), _p(x_scale), _p(w_scale), M, K, x.stride(0), hq, hkv,
        head_dim, positions.data_ptr(), cos_sin_cache.data_ptr(), _dt(cos_sin_cache), cos_sin_cache.shape[1],
        1 if is_neox else 0, k_buffer.data_ptr(), v_buffer.data_ptr(), loc.data_ptr(), _IT[loc.dtype],
        k_buffer.shape[0], _DT[dtype] | (_kvdt(qkv, k_buffer, v_buffer) & (KV_HEAD_MAJOR | KV_INTERLEAVED)), _stream(x)),
        "qkv_proj_rope_tiled")
    return qkv


def pack_weight(weight: torch.Tensor) -> torch.Tensor:
    """[N, K] 16-bit weight -> MFMA-fragment-ordered copy for linear_packed (decode GEMMs)."""
    _gpu(weight)
    assert weight.dim() == 2 and weight.is_contiguous() and weight.dtype in (torch.bfloat16, torch.float16)
    N, K = weight.shape
    lib = _lib_()
    out = torch.empty(lib.mi355_packed_weight_elems(N, K), dtype=weight.dtype, device=weight.device)
    check(lib.mi355_pack_weight(out.data_ptr(), weight.data_ptr(), N, K, _dt(weight), _stream(weight)),
          "pack_weight")
    return out


def linear_packed(x: torch.Tensor, weight_packed: torch.Tensor, n: int, bias: Optional[torch.Tensor] = None,
                  out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """linear() for M <= 64 on a weight produced by pack_weight (n = number of output features)."""
    _gpu(x, weight_packed, bias)
    assert x.dim() == 2 and x.stride(1) == 1 and weight_packed.dtype == x.dtype
    M, K = x.shape
    out_dtype = out_dtype or x.dtype
    y = torch.empty((M, n), dtype=out_dtype, device=x.device)
    if M == 0:
        return y
    lib = _lib_()
    assert weight_packed.numel() == lib.mi355_packed_weight_elems(n, K)
    need = lib.mi355_linear_workspace_bytes(M, n, K)
    ws = _workspace(x.device, need) if need > 0 else None
    check(lib.mi355_linear_packed(y.data_ptr(), x.data_ptr(), weight_packed.data_ptr(), _p(bias), M, n, K,
                                  x.stride(0), _p(ws), ws.numel() if ws is not None else 0, _dt(x),
                                  _DT[out_dtype], _stream(x)), "linear_packed")
    return y


def linear_packed_partials(x: torch.Tensor, weight_packed: torch.Tensor, n: int) -> Optional[torch.Tensor]:
    """x @ W^T left as fp32 split-K partials [splits, M, N] for fused_add_rmsnorm_partials to sum
    (None when the shape is not served: the caller then runs linear_packed)."""
    _gpu(x, weight_packed)
    assert x.dim() == 2 and x.stride(1) == 1
    m, k = x.shape
    splits = _lib_().mi355_linear_packed_splits(max(m, 1), n, k)
    if splits <= 0 or m > 64 or x.dtype not in (torch.bfloat16, torch.float16):
        return None
    part = torch.empty((splits, m, n), dtype=torch.float32, device=x.device)
    if m:
        check(_lib_().mi355_linear_packed_partials(part.data_ptr(), x.data_ptr(), weight_packed.data_ptr(), m, n, k,
                                                   x.stride(0), _dt(x), _stream(x)), "linear_packed_partials")
    return part


def fused_add_rmsnorm_partials(part: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                               eps: float) -> torch.Tensor:
    """y = sum of the split-K partials (rounded once to residual.dtype), then fused_add_rmsnorm(y,
    residual): returns the normalised rows, residual updated in place."""
    _gpu(part, residual, weight)
    assert part.dim() == 3 and part.is_contiguous() and part.dtype == torch.float32
    assert residual.is_contiguous() and residual.shape == part.shape[1:] and residual.dtype == weight.dtype
    x = torch.empty_like(residual)
    if residual.numel():
        check(_lib_().mi355_fused_add_rmsnorm_partials(x.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                                       part.data_ptr(), part.shape[0], part.shape[1], part.shape[2],
                                                       float(eps), _dt(residual), _stream(residual)),
              "fused_add_rmsnorm_partials")
    return x


def fp8_packed_partials(xq: torch.Tensor, wq_packed: torch.Tensor, n: int) -> Optional[torch.Tensor]:
    """fp8 x fp8 GEMM of fp8_scaled_mm_packed left as raw fp32 split-K partials [splits, M, N]; the
    consumer applies (sum * x_scale[m]) * w_scale[n].  None when the shape is not served."""
    _gpu(xq, wq_packed)
    assert xq.dim() == 2 and xq.is_contiguous() and xq.dtype == torch.float8_e4m3fn
    m, k = xq.shape
    splits = _lib_().mi355_fp8_packed_splits(max(m, 1), n, k)
    if splits <= 0 or m > 64:
        return None
    part = torch.empty((splits, m, n), dtype=torch.float32, device=xq.device)
    if m:
        check(_lib_().mi355_fp8_packed_partials(part.data_ptr(), xq.data_ptr(), wq_packed.data_ptr(), m, n, k,
                                                _stream(xq)), "fp8_packed_partials")
    return part


def rmsnorm_ex(weight: torch.Tensor, eps: float, x: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, part: Optional[torch.Tensor] = None,
               part_x_scale: Optional[torch.Tensor] = None, part_w_scale: Optional[torch.Tensor] = None,
               quant_fp8: bool = False):
    """General decode row norm: input x or split-K partials [splits, M, hidden] (with the fp8 GEMM's
    scales when they come from one), optional fused add of residual (updated in place), and optionally
    the per-token fp8 quantisation of the result.  Returns (out, xq or None, x_scale or None)."""
    src = part if part is not None else x
    _gpu(src, weight, residual)
    rows, hidden = (part.shape[1], part.shape[2]) if part is not None else x.shape
    dt = weight.dtype
    out = torch.empty((rows, hidden), dtype=dt, device=weight.device)
    if part is not None:
        assert part.is_contiguous() and part.dtype == torch.float32
    else:
        assert x.is_contiguous() and x.dtype == dt
    if residual is not None:
        assert residual.is_contiguous() and residual.shape == (rows, hidden) and residual.dtype == dt
    xq = xs = None
    if quant_fp8:
        xq = torch.empty((rows, hidden), dtype=torch.float8_e4m3fn, device=weight.device)
        xs = torch.empty((rows, 1), dtype=torch.float32, device=weight.device)
    if rows:
        check(_lib_().mi355_rmsnorm_ex(out.data_ptr(), _p(x), _p(residual), weight.data_ptr(), _p(part),
                                       part.shape[0] if part is not None else 0, _p(part_x_scale), _p(part_w_scale),
                                       _p(xq), _p(xs), rows, hidden, float(eps), _dt(weight), _stream(weight)),
              "rmsnorm_ex")
    return out, xq, xs


def silu_and_mul_quant_fp8(x: torch.Tensor, pad_to: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """per_token_quant_fp8(silu_and_mul(x)) in one launch: x [M, 2d] 16-bit -> (e4m3fn [M, d], scale [M, 1]);
    pad_to > d: rows of pad_to bytes with zeros behind the d values (see per_token_quant_fp8)."""
    _gpu(x)
    assert x.dim() == 2 and x.is_contiguous() and x.shape[1] % 2 == 0
    m, d = x.shape[0], x.shape[1] // 2
    qs = max(d, pad_to or 0)
    xq = torch.empty((m, qs), dtype=torch.float8_e4m3fn, device=x.device)
    xs = torch.empty((m, 1), dtype=torch.float32, device=x.device)
    if m:
        check(_lib_().mi355_silu_and_mul_quant_fp8(xq.data_ptr(), xs.data_ptr(), x.data_ptr(), m, d, qs, _dt(x),
                                                   _stream(x)), "silu_and_mul_quant_fp8")
    return xq, xs


def prefetch_next(weight_packed: torch.Tensor, m: int, n: int, k: int, budget_bytes: int = 24 << 20) -> None:
    """Hint: the next rmsnorm / fused_add_rmsnorm launch of this thread also pulls the head of this
    packed weight into L2, in the order the M-row skinny GEMM over it will ask for it (24 MiB of the
    32 MiB of L2 by default).  Results never change; shapes the hint does not serve are ignored."""
    check(_lib_().mi355_prefetch_next(weight_packed.data_ptr(), m, n, k, weight_packed.element_size(),
                                      budget_bytes), "prefetch_next")


def pack_weight_gate_up(weight: torch.Tensor) -> torch.Tensor:
    """[2d, K] = [gate rows | up rows] -> packed copy for linear_packed_silu_mul: every 16-row MFMA
    tile holds 8 gate rows followed by the 8 up rows of the same output features."""
    n2, K = weight.shape
    d = n2 // 2
    assert n2 % 16 == 0
    inter = torch.stack([weight[:d].view(d // 8, 8, K), weight[d:].view(d // 8, 8, K)], dim=1)
    return pack_weight(inter.reshape(n2, K).contiguous())


def linear_packed_silu_mul(x: torch.Tensor, weight_packed: torch.Tensor, n: int) -> torch.Tensor:
    """SiluAndMul(x @ W_gate_up^T) in one kernel (models/llama.py:94-98) for M <= 64 rows;
    weight_packed from pack_weight_gate_up, n = 2d rows, returns [M, d]."""
    _gpu(x, weight_packed)
    assert x.dim() == 2 and x.stride(1) == 1 and weight_packed.dtype == x.dtype
    M, K = x.shape
    y = torch.empty((M, n // 2), dtype=x.dtype, device=x.device)
    if M == 0:
        return y
    lib = _lib_()
    assert weight_packed.numel() == lib.mi355_packed_weight_elems(n, K)
    need = lib.mi355_linear_workspace_bytes(M, n, K)
    ws = _workspace(x.device, need) if need > 0 else None
    check(lib.mi355_linear_packed_silu_mul(y.data_ptr(), x.data_ptr(), weight_packed.data_ptr(), M, n, K,
                                           x.stride(0), _p(ws), ws.numel() if ws is not None else 0,
                                           _dt(x), _stream(x)), "linear_packed_silu_mul")
    return y


def per_token_quant_fp8(x: torch.Tensor, pad_to: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """per_token_group_quant_fp8(x, group_size=K) (fp8_kernel.py:115-155) -> (xq e4m3fn, scale [M,1]).
    pad_to > K: xq is [M, pad_to] with zeros behind the K values of every row -- the activation of a packed GEMM
    whose reduction length was rounded up with zero weights."""
    _gpu(x)
    assert x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    qs = max(K, pad_to or 0)
    xq = torch.empty((M, qs), dtype=torch.float8_e4m3fn, device=x.device)
    s = torch.empty((M, 1), dtype=torch.float32, device=x.device)
    check(_lib_().mi355_per_token_quant_fp8(xq.data_ptr(), s.data_ptr(), x.data_ptr(), M, K,
                                            x.stride(0), qs, _dt(x), _stream(x)), "per_token_quant_fp8")
    return xq, s


def fp8_scaled_mm(xq: torch.Tensor, wq: torch.Tensor, x_scale: torch.Tensor, w_scale: torch.Tensor,
                  out_dtype: torch.dtype, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias) (sgl-kernel gemm.py) with
    wq given as [N, K] (one row per output channel)."""
    _gpu(xq, wq, x_scale, w_scale, bias)
    assert xq.is_contiguous() and wq.is_contiguous()
    M, K = xq.shape
    N = wq.shape[0]
    assert wq.shape[1] == K and x_scale.numel() == M and w_scale.numel() == N
    assert x_scale.dtype == torch.float32 and w_scale.dtype == torch.float32
    assert bias is None or bias.dtype == out_dtype
    y = torch.empty((M, N), dtype=out_dtype, device=xq.device)
    lib = _lib_()
    need = lib.mi355_fp8_workspace_bytes(M, N, K)
    ws = _workspace(xq.device, need, "fp8") if need > 0 else None
    check(lib.mi355_fp8_scaled_mm(y.data_ptr(), xq.data_ptr(), wq.data_ptr(), x_scale.data_ptr(),
                                  w_scale.data_ptr(), _p(bias), M, N, K, _p(ws),
                                  ws.numel() if ws is not None else 0, _DT[out_dtype],
                                  _stream(xq)), "fp8_scaled_mm")
    return y


def fp8_scaled_mm_tiled_supported(m: int, n: int, k: int, out_dtype: torch.dtype, silu_mul: bool = False) -> bool:
    """Whether fp8_scaled_mm_tiled serves the shape (16-bit output, K % 128 == 0, K >= 256, N % 16 == 0)."""
    return out_dtype in (torch.bfloat16, torch.float16) and bool(
        _lib_().mi355_fp8_scaled_mm_tiled_supported(m, n, k, _DT[out_dtype], 1 if silu_mul else 0))


def fp8_scaled_mm_tiled(xq: torch.Tensor, wq: torch.Tensor, x_scale: torch.Tensor, w_scale: torch.Tensor,
                        out_dtype: torch.dtype, bias: Optional[torch.Tensor] = None,
                        silu_mul: bool = False) -> torch.Tensor:
    """fp8_scaled_mm at prefill sizes on the 256 x 256-tile matrix-core kernel (wq [N, K]);
    silu_mul=True applies SiluAndMul to the [gate; up] output in the epilogue and returns [M, N / 2]."""
    _gpu(xq, wq, x_scale, w_scale, bias)
    assert xq.is_contiguous() and wq.is_contiguous()
    assert xq.dtype == torch.float8_e4m3fn and wq.dtype == torch.float8_e4m3fn
    M, K = xq.shape
    N = wq.shape[0]
    assert wq.shape[1] == K and x_scale.numel() == M and w_scale.numel() == N
    assert x_scale.dtype == torch.float32 and w_scale.dtype == torch.float32
    assert x_scale.is_contiguous() and w_scale.is_contiguous()
    assert bias is None or bias.dtype == out_dtype
    y = torch.empty((M, N // 2 if silu_mul else N), dtype=out_dtype, device=xq.device)
    if M == 0:
        return y
    check(_lib_().mi355_fp8_scaled_mm_tiled(y.data_ptr(), xq.data_ptr(), wq.data_ptr(), x_scale.data_ptr(),
                                            w_scale.data_ptr(), _p(bias), M, N, K, _DT[out_dtype],
                                            1 if silu_mul else 0, _stream(xq)), "fp8_scaled_mm_tiled")
    return y


def pack_weight_fp8(wq: torch.Tensor) -> torch.Tensor:
    """[N, K] e4m3fn weight -> packed copy for fp8_scaled_mm_packed (same 16-byte-fragment layout
    as pack_weight: the bytes are packed as an [N, K/2] 16-bit matrix)."""
    assert wq.dtype == torch.float8_e4m3fn and wq.is_contiguous() and wq.shape[1] % 64 == 0
    return pack_weight(wq.view(torch.bfloat16)).view(torch.float8_e4m3fn)


def fp8_scaled_mm_packed(xq: torch.Tensor, wq_packed: torch.Tensor, n: int, x_scale: torch.Tensor,
                         w_scale: torch.Tensor, out_dtype: torch.dtype,
                         bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp8_scaled_mm for M <= 64 on a weight from pack_weight_fp8 (n output features)."""
    _gpu(xq, wq_packed, x_scale, w_scale, bias)
    M, K = xq.shape
    assert xq.is_contiguous() and xq.dtype == torch.float8_e4m3fn and wq_packed.dtype == torch.float8_e4m3fn
    assert x_scale.dtype == torch.float32 and x_scale.numel() == M and w_scale.dtype == torch.float32
    assert w_scale.numel() == n and (bias is None or bias.dtype == out_dtype)
    y = torch.empty((M, n), dtype=out_dtype, device=xq.device)
    if M == 0:
        return y
    lib = _lib_()
    need = lib.mi355_fp8_packed_workspace_bytes(M, n, K)
    ws = _workspace(xq.device, need) if need > 0 else None
    check(lib.mi355_fp8_scaled_mm_packed(y.data_ptr(), xq.data_ptr(), wq_packed.data_ptr(), x_scale.data_ptr(),
                                         w_scale.data_ptr(), _p(bias), M, n, K, _p(ws),
                                         ws.numel() if ws is not None else 0, _DT[out_dtype], _stream(xq)),
          "fp8_scaled_mm_packed")
    return y


# ---------------------------------------------------------------------------------- sampling
def argmax(logits: torch.Tensor) -> torch.Tensor:
    """torch.argmax(logits, -1) on fp32 [B, V] (layers/sampler.py:74)."""
    _gpu(logits)
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1
    ids = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    rows = logits.shape[0]
    if 0 < rows < 256:
        # few rows: several workgroups per row (mi355_argmax_chunked); its cells live in a persistent zeroed workspace
        lib = _lib_()
        need = lib.mi355_argmax_workspace_bytes(rows)
        ws = _workspace(logits.device, need, tag="argmax")
        check(lib.mi355_argmax_chunked(ids.data_ptr(), logits.data_ptr(), rows, logits.shape[1], logits.stride(0),
                                       ws.data_ptr(), ws.numel(), _stream(logits)), "argmax_chunked")
        return ids
    check(_lib_().mi355_argmax(ids.data_ptr(), logits.data_ptr(), logits.shape[0], logits.shape[1],
                               logits.stride(0), _stream(logits)), "argmax")
    return ids


def softmax_temperature_(logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
    """In place logits.div_(temperatures); logits[:] = softmax(logits) (layers/sampler.py:79-81)."""
    _gpu(logits, temperatures)
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1
    assert temperatures.dtype == torch.float32 and temperatures.numel() == logits.shape[0]
    check(_lib_().mi355_softmax_temperature(logits.data_ptr(), temperatures.data_ptr(),
                                            logits.shape[0], logits.shape[1], logits.stride(0),
                                            _stream(logits)), "softmax_temperature")
    return logits


def top_k_top_p_min_p_sample(probs: torch.Tensor, top_ks: Optional[torch.Tensor],
                             top_ps: Optional[torch.Tensor], min_ps: Optional[torch.Tensor],
                             uniform: torch.Tensor) -> torch.Tensor:
    """Joint top-k/top-p/min-p sampling from probabilities with caller-supplied uniforms."""
    _gpu(probs, top_ks, top_ps, min_ps, uniform)
    assert probs.dim() == 2 and probs.dtype == torch.float32 and probs.stride(1) == 1
    B = probs.shape[0]
    assert uniform.dtype == torch.float32 and uniform.numel() == B
    assert top_ks is None or (top_ks.dtype == torch.int32 and top_ks.numel() == B)
    assert top_ps is None or (top_ps.dtype == torch.float32 and top_ps.numel() == B)
    assert min_ps is None or (min_ps.dtype == torch.float32 and min_ps.numel() == B)
    ids = torch.empty(B, dtype=torch.int32, device=probs.device)
    check(_lib_().mi355_top_k_top_p_min_p_sample(ids.data_ptr(), probs.data_ptr(), _p(top_ks),
                                                 _p(top_ps), _p(min_ps), uniform.data_ptr(), B,
                                                 probs.shape[1], probs.stride(0), _stream(probs)),
          "top_k_top_p_min_p_sample")
    return ids
