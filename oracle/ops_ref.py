"""Per-op CPU restatements (torch CPU tensors; any float dtype).  TEST INFRASTRUCTURE."""
import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


# ---- layers/layernorm.py:128-147 RMSNorm.forward_native ------------------------------------
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float,
            residual: Optional[torch.Tensor] = None):
    orig = x.dtype
    x = x.to(torch.float32)
    if residual is not None:
        x = x + residual.to(torch.float32)
        residual = x.to(orig)
    variance = x.pow(2).mean(dim=-1, keepdim=True)
    x = x * torch.rsqrt(variance + eps)
    x = (x * weight).to(orig)
    return x if residual is None else (x, residual)


# ---- layers/activation.py:55-58 SiluAndMul.forward_native -----------------------------------
def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    d = x.shape[-1] // 2
    return F.silu(x[..., :d]) * x[..., d:]


# ---- layers/logits_processor.py:579-604 fused_softcap_kernel ----------------------------------
# The reference's kernel is Triton (no CPU build of it exists): this is its arithmetic restated in
# fp32, x / cap -> (e^{2x} - 1) / (e^{2x} + 1) -> * cap; parity unpinned by reference outputs, the
# tests also hold it to torch.tanh in fp64.  Where e^{2x} overflows fp32 (|x / cap| > 44) the
# reference's quotient is inf / inf; this restatement returns the limit, +cap.
def softcap(x: torch.Tensor, cap: float) -> torch.Tensor:
    y = x.float() / cap
    e = torch.exp(torch.clamp(2 * y, max=40.0))
    return (e - 1) / (e + 1) * cap


# ---- layers/rotary_embedding.py:113-136 cache, :806-827 llama3 inverse frequencies -----------
def rope_inv_freq(rotary_dim: int, base: float, llama3: Optional[dict] = None) -> torch.Tensor:
    inv = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    if llama3 is None:
        return inv
    orig = llama3["original_max_position_embeddings"]
    lo_f, hi_f, factor = llama3["low_freq_factor"], llama3["high_freq_factor"], llama3["factor"]
    low_wavelen, high_wavelen = orig / lo_f, orig / hi_f
    wave_len = 2 * math.pi / inv
    smooth = (orig / wave_len - lo_f) / (hi_f - lo_f) if lo_f != hi_f else 0
    return torch.where(wave_len < high_wavelen, inv,
                       torch.where(wave_len > low_wavelen, inv / factor,
                                   (1 - smooth) * inv / factor + smooth * inv))


def cos_sin_cache(rotary_dim: int, max_pos: int, base: float, llama3: Optional[dict] = None,
                  dtype: torch.dtype = torch.float32) -> torch.Tensor:
    t = torch.arange(max_pos, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, rope_inv_freq(rotary_dim, base, llama3))
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype)


# ---- layers/rotary_embedding.py:49-75 _apply_rotary_emb, :138-166 forward_native ------------
def _rotate(x, cos, sin, is_neox):
    cos = cos.unsqueeze(-2).to(x.dtype)
    sin = sin.unsqueeze(-2).to(x.dtype)
    if is_neox:
        x1, x2 = torch.chunk(x, 2, dim=-1)
    else:
        x1, x2 = x[..., ::2], x[..., 1::2]
    o1 = x1 * cos - x2 * sin
    o2 = x2 * cos + x1 * sin
    if is_neox:
        return torch.cat((o1, o2), dim=-1)
    return torch.stack((o1, o2), dim=-1).flatten(-2)


def apply_rope(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
               cache: torch.Tensor, is_neox: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    positions = positions.flatten()
    n = positions.shape[0]
    rotary_dim = cache.shape[1]
    cos, sin = cache.index_select(0, positions).chunk(2, dim=-1)

    def one(x):
        shape = x.shape
        x = x.reshape(n, -1, head_size)
        rot = _rotate(x[..., :rotary_dim], cos, sin, is_neox)
        return torch.cat((rot, x[..., rotary_dim:]), dim=-1).reshape(shape)

    return one(query), one(key)


# ---- layers/attention/torch_native_backend.py:112-180 decode, :27-110 extend ----------------
def decode_attention(q, k_cache, v_cache, req_to_token, req_pool_indices, seq_lens, scaling,
                     enable_gqa: bool) -> torch.Tensor:
    """q [B, Hq, D]; caches [slots, Hkv, D]; returns [B, Hq, Dv]."""
    out = torch.empty((q.shape[0], q.shape[1], v_cache.shape[-1]), dtype=q.dtype)
    qh = q.movedim(0, 1)  # [Hq, B, D]
    for i in range(seq_lens.shape[0]):
        toks = req_to_token[req_pool_indices[i], : seq_lens[i]].long()
        k = k_cache[toks].movedim(0, 1)
        v = v_cache[toks].movedim(0, 1)
        o = F.scaled_dot_product_attention(qh[:, i:i + 1].unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0),
                                           enable_gqa=enable_gqa, scale=scaling, is_causal=False)
        out[i] = o.squeeze(0).movedim(1, 0)[0]
    return out


def extend_attention(q, k_cache, v_cache, req_to_token, req_pool_indices, seq_lens,
                     extend_prefix_lens, extend_seq_lens, scaling, enable_gqa: bool,
                     causal: bool = True) -> torch.Tensor:
    """q [T, Hq, D] (new tokens, all sequences concatenated); K/V of the new tokens must already
    be in the caches.  The reference pads q with `prefix` junk rows so that is_causal lines up
    (:81-87); only the real rows are kept."""
    out = torch.empty((q.shape[0], q.shape[1], v_cache.shape[-1]), dtype=q.dtype)
    qh = q.movedim(0, 1)
    start = 0
    for i in range(seq_lens.shape[0]):
        ext, pre, total = int(extend_seq_lens[i]), int(extend_prefix_lens[i]), int(seq_lens[i])
        qq = torch.zeros((qh.shape[0], total, qh.shape[2]), dtype=q.dtype)
        qq[:, pre:, :] = qh[:, start:start + ext]
        toks = req_to_token[req_pool_indices[i], :total].long()
        k = k_cache[toks].movedim(0, 1)
        v = v_cache[toks].movedim(0, 1)
        o = F.scaled_dot_product_attention(qq.unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0),
                                           enable_gqa=enable_gqa, scale=scaling, is_causal=causal)
        out[start:start + ext] = o.squeeze(0).movedim(1, 0)[pre:]
        start += ext
    return out


def attention_fp64(q, k, v, scaling, causal_offset: Optional[int] = None, logit_cap: float = 0.0):
    """Plain softmax(q k^T) v in float64 for shapes/features SDPA does not cover (logit cap,
    custom masks): q [Hq, Lq, D], k/v [Hkv, S, D*]; row i sees keys <= i + causal_offset."""
    g = q.shape[0] // k.shape[0]
    k = k.repeat_interleave(g, 0).double()
    v = v.repeat_interleave(g, 0).double()
    s = torch.einsum("hqd,hkd->hqk", q.double(), k) * scaling
    if logit_cap > 0:
        s = logit_cap * torch.tanh(s / logit_cap)
    if causal_offset is not None:
        S, Lq = k.shape[1], q.shape[1]
        keep = torch.arange(S)[None, :] <= torch.arange(Lq)[:, None] + causal_offset
        s = s.masked_fill(~keep, float("-inf"))
    return torch.einsum("hqk,hkd->hqd", torch.softmax(s, -1), v)


# ---- layers/attention/utils.py:5-41, managers/schedule_batch.py:1287-1294,1571-1573 ----------
def window_of(length: int, sliding_window_size: int):
    """(start, count) of the cached tokens a sliding-window layer is shown: the last
    min(length, window + 1) (update_sliding_window_buffer, layers/attention/triton_backend.py:927-955)."""
    n = min(int(length), sliding_window_size + 1)
    return int(length) - n, n


def extend_attention_kernel_rules_fp64(q, k_prefix, v_prefix, k_ext, v_ext, scaling, causal=True,
                                       logit_cap: float = 0.0, sliding_window_size: int = -1):
    """The masking rules of the reference's extend kernel (triton_ops/extend_attention.py:41-303) for
    ONE sequence in float64: q [E, Hq, D]; k_prefix / v_prefix [P, Hkv, *] are the cached keys as the
    kernel is handed them (for a sliding-window layer: already cut to window_of(prefix)); k_ext / v_ext
    [E, Hkv, *] the new tokens.  Phase 1 (prefix) keeps key k for query row i iff i <= k + window when a
    window is set (:179-184; i counts new tokens, k counts the handed prefix keys); phase 2 (new
    tokens) is causal and has no window term (:247-262).  logit_cap * tanh(s / logit_cap) on both.
    Parity unpinned: the reference's kernel is Triton (no CPU execution) and its torch-native backend
    ignores windows; this follows the kernel's source."""
    E, Hq, _ = q.shape
    P = k_prefix.shape[0]
    g = Hq // k_ext.shape[1]
    k = torch.cat([k_prefix, k_ext]).repeat_interleave(g, 1).double()      # [P + E, Hq, D]
    v = torch.cat([v_prefix, v_ext]).repeat_interleave(g, 1).double()
    s = torch.einsum("ehd,khd->hek", q.double(), k) * scaling
    if logit_cap > 0:
        s = logit_cap * torch.tanh(s / logit_cap)
    keep = torch.ones(E, P + E, dtype=torch.bool)
    i = torch.arange(E)[:, None]
    if sliding_window_size > 0 and P:
        keep[:, :P] = i <= torch.arange(P)[None, :] + sliding_window_size
    if causal:
        keep[:, P:] = i >= torch.arange(E)[None, :]
    s = s.masked_fill(~keep[None], float("-inf"))
    return torch.einsum("hek,khd->ehd", torch.softmax(s, -1), v)


def create_kv_indices(req_to_token, req_pool_indices, lens, kv_start=None) -> torch.Tensor:
    parts = []
    for i in range(len(req_pool_indices)):
        s = 0 if kv_start is None else int(kv_start[i])
        parts.append(req_to_token[req_pool_indices[i], s: s + int(lens[i])])
    return torch.cat(parts) if parts else torch.empty(0, dtype=torch.int32)


def write_req_to_token(req_to_token, req_pool_indices, prefix_lens, seq_lens, out_cache_loc):
    pt = 0
    for i in range(len(req_pool_indices)):
        n = int(seq_lens[i]) - int(prefix_lens[i])
        req_to_token[req_pool_indices[i], int(prefix_lens[i]): int(seq_lens[i])] = out_cache_loc[pt: pt + n].to(torch.int32)
        pt += n


# ---- model_executor/forward_batch_info.py:759-778 ---------------------------------------------
def compute_position(extend_prefix_lens, extend_seq_lens):
    positions = torch.cat([torch.arange(int(p), int(p) + int(e)) for p, e in zip(extend_prefix_lens, extend_seq_lens)])
    start = torch.zeros_like(extend_seq_lens)
    start[1:] = torch.cumsum(extend_seq_lens[:-1], dim=0)
    return positions.to(torch.int64), start


def clamp_position(seq_lens):
    return torch.clamp(seq_lens - 1, min=0).to(torch.int64)


# ---- layers/sampler.py:72-82, 154-178 ------------------------------------------------------------
def greedy(logits: torch.Tensor) -> torch.Tensor:
    return torch.argmax(logits, -1)


def softmax_temperature(logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
    return torch.softmax(logits / temperatures.view(-1, 1), dim=-1)


def top_k_top_p_min_p_filter(probs, top_ks, top_ps, min_ps=None):
    """The masked, SORTED probabilities and their vocabulary indices exactly as
    top_k_top_p_min_p_sampling_from_probs_torch builds them before torch.multinomial."""
    probs_sort, probs_idx = probs.sort(dim=-1, descending=True)
    probs_sum = torch.cumsum(probs_sort, dim=-1)
    probs_sort[torch.arange(0, probs.shape[-1]).view(1, -1) >= top_ks.view(-1, 1)] = 0.0
    probs_sort[(probs_sum - probs_sort) > top_ps.view(-1, 1)] = 0.0
    if min_ps is not None:
        thr = probs_sort[:, 0] * min_ps
        probs_sort[probs_sort < thr.view(-1, 1)] = 0.0
    return probs_sort, probs_idx


def top_k_top_p_min_p_sampling_from_probs(probs, top_ks, top_ps, min_ps=None):
    probs_sort, probs_idx = top_k_top_p_min_p_filter(probs.clone(), top_ks, top_ps, min_ps)
    sampled = torch.multinomial(probs_sort, num_samples=1)
    return torch.gather(probs_idx.to(torch.int32), dim=1, index=sampled).view(-1)


def kept_distribution(probs, top_ks, top_ps, min_ps=None):
    """Renormalised distribution over the vocabulary that the reference sampler draws from."""
    ps, pi = top_k_top_p_min_p_filter(probs.clone(), top_ks, top_ps, min_ps)
    ps = ps / ps.sum(-1, keepdim=True)
    return torch.zeros_like(ps).scatter_(-1, pi, ps)


# ---- layers/quantization/fp8_kernel.py:144-155, fp8_utils.py:479-507 ---------------------------
FP8_MAX = 448.0  # OCP e4m3fn (gfx950); fp8_kernel.py:71-85 picks fnuz only on gfx94x


def per_token_quant_fp8(x: torch.Tensor):
    y = x.to(torch.float32)
    absmax = torch.clamp(y.abs().amax(dim=-1, keepdim=True), min=1e-10)
    y_s = absmax / FP8_MAX
    y_q = torch.clamp(y * (1.0 / y_s), -FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return y_q, y_s


def fp8_linear(x: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor, bias=None) -> torch.Tensor:
    """wq [N, K] fp8, w_scale [N, 1] fp32: out = ((xq @ wq^T) * x_scale * w_scale^T (+bias)).to(x.dtype)."""
    xq, xs = per_token_quant_fp8(x)
    out = xq.to(torch.float32) @ wq.to(torch.float32).t()
    out = out * xs * w_scale.t()
    if bias is not None:
        out = out + bias
    return out.to(x.dtype)


def alloc_extend_pages(prefix_lens, seq_lens, last_loc, free_pages, page_size: int):
    """Page-aligned slots for an extend batch, request by request
    (python/sglang/srt/mem_cache/allocator.py:276-360; same arithmetic as the torch variant :573-625):
    (1) the rest of the request's last page, (2) whole new pages, (3) the head of one more new page;
    new pages are taken from the front of free_pages in request order.
    Returns (out_indices int64 [sum extend], pages_used)."""
    out, used = [], 0
    for pre, seq, ll in zip([int(v) for v in prefix_lens], [int(v) for v in seq_lens], [int(v) for v in last_loc]):
        pre_up = (pre + page_size - 1) // page_size * page_size
        n1 = min(seq, pre_up) - pre
        out += [ll + 1 + i for i in range(n1)]
        n_new = (seq + page_size - 1) // page_size - (pre + page_size - 1) // page_size
        mine = [int(free_pages[used + i]) for i in range(n_new)]
        if pre + n1 < seq:
            n2 = seq // page_size * page_size - pre_up
            out += [mine[i // page_size] * page_size + i % page_size for i in range(n2)]
            if pre + n1 + n2 < seq:
                n3 = seq - seq // page_size * page_size
                out += [mine[-1] * page_size + i for i in range(n3)]
        used += n_new
    return torch.tensor(out, dtype=torch.int64), used


def alloc_decode_pages(seq_lens, last_loc, free_pages, page_size: int):
    """One new slot per request for a decode step (allocator.py:364-401): the slot after last_loc,
    or the first slot of a new page when seq_len - 1 fills whole pages.  seq_lens counts the new token."""
    out, used = [], 0
    for seq, ll in zip([int(v) for v in seq_lens], [int(v) for v in last_loc]):
        if (seq + page_size - 1) // page_size - (seq - 1 + page_size - 1) // page_size:
            out.append(int(free_pages[used]) * page_size)
            used += 1
        else:
            out.append(ll + 1)
    return torch.tensor(out, dtype=torch.int64), used


def kv_to_fp8_and_back(x: torch.Tensor) -> torch.Tensor:
    """What an fp8 KV cache (kv_cache_dtype fp8_e4m3, mem_cache/memory_pool.py:436-446) hands to
    the attention kernels: x cast to e4m3fn (round to nearest even; values beyond +-448 saturate)
    and widened back to x.dtype (the Triton kernels cast K / V to the query dtype before the dot)."""
    return x.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(x.dtype)



def custom_all_reduce(inputs):
    """Sum of the ranks' tensors as the reference's one-stage kernel forms it: upcast to fp32,
    accumulate in rank order 0..n-1, round once to the tensor's dtype
    (sgl-kernel/csrc/allreduce/custom_all_reduce_hip.cuh:250-258 packed_reduce, :273-279)."""
    acc = inputs[0].float()
    for t in inputs[1:]:
        acc = acc + t.float()
    return acc.to(inputs[0].dtype)
