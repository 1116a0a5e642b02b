"""CPU checker implementation of the ops surface the host-side layers call (tests only).

Installed with sglang_custom_amd.ops_backend.set_ops_backend() so that the batch plumbing, the
model wiring and tensor-parallel sharding can be exercised without a GPU; every function is the
oracle's restatement of the reference op, adapted to the calling convention of
sglang_custom_amd.ops (pre-allocated / in-place outputs)."""
import torch
import torch.nn.functional as F

from oracle import ops_ref


def rmsnorm(x, weight, eps, out=None):
    y = ops_ref.rmsnorm(x, weight, eps)
    if out is None:
        return y
    out.copy_(y)
    return out


def fused_add_rmsnorm(x, residual, weight, eps):
    y, r = ops_ref.rmsnorm(x, weight, eps, residual)
    x.copy_(y)
    residual.copy_(r)


def silu_and_mul(x, out=None):
    y = ops_ref.silu_and_mul(x)
    if out is None:
        return y
    out.copy_(y)
    return out


def softcap_inplace(logits, cap):
    logits.copy_(ops_ref.softcap(logits, cap))
    return logits


def rope_inplace(positions, q, k, head_size, cos_sin_cache, is_neox, v=None, k_buffer=None,
                 v_buffer=None, loc=None):
    qn, kn = ops_ref.apply_rope(positions, q, k, head_size, cos_sin_cache, is_neox)
    q.copy_(qn)
    k.copy_(kn)
    if k_buffer is not None:
        T = positions.numel()
        k_buffer[loc.long()] = k.reshape(T, -1, head_size)
        v_buffer[loc.long()] = v.reshape(T, -1, v_buffer.shape[-1])


def set_kv_buffer(k_buffer, v_buffer, loc, cache_k, cache_v):
    T = loc.numel()
    k_buffer[loc.long()] = cache_k.reshape(T, *k_buffer.shape[1:])
    v_buffer[loc.long()] = cache_v.reshape(T, *v_buffer.shape[1:])


def embedding(ids, table, vocab_start=0, vocab_end=None):
    if vocab_end is None:
        vocab_end = vocab_start + table.shape[0]
    ok = (ids >= vocab_start) & (ids < vocab_end)
    out = table[(ids - vocab_start).clamp(0, table.shape[0] - 1)]
    out = out * ok.unsqueeze(-1).to(out.dtype)
    return out


def create_kv_indices(req_to_token, req_pool_indices, page_kernel_lens, kv_indptr, kv_start_idx, kv_indices):
    r = ops_ref.create_kv_indices(req_to_token, req_pool_indices, page_kernel_lens, kv_start_idx)
    kv_indices[: r.numel()] = r


def write_req_to_token(req_to_token, req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc):
    ops_ref.write_req_to_token(req_to_token, req_pool_indices, pre_lens, seq_lens, out_cache_loc)


def compute_position(extend_prefix_lens, extend_seq_lens, extend_seq_lens_sum):
    pos, start = ops_ref.compute_position(extend_prefix_lens, extend_seq_lens)
    return pos, start.to(torch.int32)


def prepare_decode(req_to_token, req_pool_indices, seq_lens, out_cache_loc, positions):
    seq_lens.add_(1)
    positions.copy_(ops_ref.clamp_position(seq_lens))
    req_to_token[req_pool_indices, seq_lens - 1] = out_cache_loc.to(torch.int32)


def decode_num_splits(bs, hkv, max_seq_len, max_splits):
    return 1


def decode_attention_paged(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens,
                           attn_logits, attn_lse, num_kv_splits, sm_scale, logit_cap=0.0, tickets=None, kv_chunk_keys=None):
    assert logit_cap == 0.0
    gqa = q.shape[1] != k_buffer.shape[1]
    o.copy_(ops_ref.decode_attention(q, k_buffer, v_buffer, req_to_token, req_pool_indices, seq_lens,
                                     sm_scale, gqa))


def decode_attention_mla(q, kv_buffer, o, req_to_token, req_pool_indices, seq_lens, attn_logits, attn_lse,
                         num_kv_splits, kv_lora_rank, sm_scale, logit_cap=0.0, tickets=None, kv_chunk_keys=None):
    """Multi-query attention over latent rows: keys = whole rows, values = their first kv_lora_rank features."""
    assert not logit_cap
    o.copy_(ops_ref.decode_attention(q, kv_buffer, kv_buffer[..., :kv_lora_rank], req_to_token, req_pool_indices,
                                     seq_lens, sm_scale, True))


def extend_attention_fwd(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr,
                         kv_indptr, kv_indices, custom_mask, is_causal, mask_indptr, max_len_extend,
                         sm_scale=None, logit_cap=0.0, skip_prefix_custom_mask=True,
                         sliding_window_size=-1):
    """Prefix keys come from kv_indices, new keys from k_extend/v_extend (extend_attention.py:306-438)."""
    assert custom_mask is None and logit_cap == 0.0 and sliding_window_size in (-1, None)
    Hq, Hkv = q_extend.shape[1], k_extend.shape[1]
    for i in range(qo_indptr.numel() - 1):
        qs, qe = int(qo_indptr[i]), int(qo_indptr[i + 1])
        ps, pe = int(kv_indptr[i]), int(kv_indptr[i + 1])
        idx = kv_indices[ps:pe].long()
        if pe > ps:
            k = torch.cat([k_buffer[idx], k_extend[qs:qe]], 0).movedim(0, 1)
            v = torch.cat([v_buffer[idx], v_extend[qs:qe]], 0).movedim(0, 1)
        else:   # no prefix: the buffers are not read (an MLA pool's rows do not even have this shape)
            k, v = k_extend[qs:qe].movedim(0, 1), v_extend[qs:qe].movedim(0, 1)
        ext, pre = qe - qs, pe - ps
        qq = torch.zeros((Hq, pre + ext, q_extend.shape[2]), dtype=q_extend.dtype)
        qq[:, pre:] = q_extend[qs:qe].movedim(0, 1)
        o = F.scaled_dot_product_attention(qq[None], k[None], v[None], enable_gqa=Hq != Hkv,
                                           scale=sm_scale, is_causal=is_causal)
        o_extend[qs:qe] = o[0].movedim(1, 0)[pre:]


def linear(x, weight, bias=None, out_dtype=None):
    y = F.linear(x, weight, bias)
    return y if out_dtype is None else y.to(out_dtype)


def argmax(logits):
    return ops_ref.greedy(logits)


def softmax_temperature_(logits, temperatures):
    logits.copy_(ops_ref.softmax_temperature(logits, temperatures.view(-1, 1)))
    return logits


def top_k_top_p_min_p_sample(probs, top_ks, top_ps, min_ps, uniform):
    kept = ops_ref.kept_distribution(probs, top_ks, top_ps, min_ps)
    cdf = kept.cumsum(-1)
    return (cdf < uniform.view(-1, 1) * cdf[:, -1:]).sum(-1).clamp(max=probs.shape[1] - 1).to(torch.int32)


def alloc_extend_pages(prefix_lens, seq_lens, last_loc, free_pages, page_size, extend_num_tokens):
    out, used = ops_ref.alloc_extend_pages(prefix_lens, seq_lens, last_loc, free_pages, page_size)
    assert out.numel() == extend_num_tokens
    return out, used


def alloc_decode_pages(seq_lens, last_loc, free_pages, page_size):
    return ops_ref.alloc_decode_pages(seq_lens, last_loc, free_pages, page_size)


def qkv_proj_rope_tiled_supported(m, k, num_heads, num_kv_heads, head_dim, rotary_dim, is_neox, dtype, fp8):
    """The fused prefill projection + RoPE + cache write is a GPU kernel only: the double declines and the
    model takes the unfused sequence, which is what the oracle restates."""
    return False


def decode_attention_mla_fused_rope_supported(kv_lora_rank, rope_dim, dtype, cache_dtype):
    """RoPE + latent-row write inside the MLA decode launch is a GPU kernel only: the double declines and the
    model takes rotary_emb + set_kv_buffer + decode_attention_mla, which is what the oracle restates."""
    return False


def mla_latent_norms_supported(q_rank, kv_rank, rope_dim, dtype):
    """The one-launch q_a / kv_a norms are a GPU kernel only: the double declines and the model takes
    split + contiguous + rmsnorm, which is what the oracle restates."""
    return False


def mla_absorb_supported(m, n, k, tail, dtype):
    """The one-launch absorbed products are a GPU kernel only: the double declines and the model takes
    torch.bmm + cat / transpose, which is what the oracle restates."""
    return False


def mla_prefill_assemble_supported(nope, rope, vd, rank, dtype):
    """The one-launch prefill glue is a GPU kernel only: the double declines and the model takes rotary_emb + the
    copies + set_kv_buffer, which is what the oracle restates."""
    return False
