"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

CPU restatement of DeepSeek-V2/V3 multi-head latent attention as the reference computes it WITHOUT
weight absorption (python/sglang/srt/models/deepseek_v2.py forward_normal_prepare/_core :1131-1179):
q = q_proj(h) (or q_b_proj(rmsnorm(q_a))), latent = kv_a_proj_with_mqa(h), c_kv = rmsnorm(latent[:rank]),
[k_nope | v] = kv_b_proj(c_kv) per head, k_pe = rope(latent[rank:]) shared by all heads, q_pe = rope(q[nope:]),
causal softmax((q . k) * scale) v, o_proj.  The rotation is the interleaved (GPT-J) one
(layers/rotary_embedding.py:78-110 _apply_rotary_emb with is_neox_style=False).  Everything in float64:
the absorbed decode path (w_kc / w_vc folded into query and output, :1181-1335) is algebraically the
same function, which is what the GPU test checks against this file.
Parity pin: the formulas above are the reference's statements line by line; there is no golden vector
for this module in the reference's tests, so the MLA module's parity is "unpinned" beyond this algebra.
"""
import torch


def _rms(x, w, eps):
    return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps) * w


def _rope_gptj(x, positions, theta):
    """x [T, H, R]; pairs (2i, 2i+1) rotate by positions * theta^(-2i/R)."""
    R = x.shape[-1]
    inv = 1.0 / (theta ** (torch.arange(0, R, 2, dtype=torch.float64) / R))
    ang = positions.double()[:, None] * inv[None]
    cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]
    x1, x2 = x[..., 0::2], x[..., 1::2]
    return torch.stack([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1).flatten(-2)


def mla_attention(hidden, w, num_heads, nope, rope, v_dim, q_lora_rank, kv_lora_rank, theta=10000.0, eps=1e-6):
    """hidden [S, hidden] of ONE sequence at positions 0..S-1 -> [S, hidden], causal."""
    h = hidden.double()
    W = {k: v.double() for k, v in w.items()}
    S = h.shape[0]
    pos = torch.arange(S)
    if q_lora_rank is not None:
        a = h @ W["fused_qkv_a_proj_with_mqa"].t()
        q = _rms(a[:, :q_lora_rank], W["q_a_layernorm"], eps) @ W["q_b_proj"].t()
        latent = a[:, q_lora_rank:]
    else:
        q = h @ W["q_proj"].t()
        latent = h @ W["kv_a_proj_with_mqa"].t()
    q = q.view(S, num_heads, nope + rope)
    c_kv = _rms(latent[:, :kv_lora_rank], W["kv_a_layernorm"], eps)
    kv = (c_kv @ W["kv_b_proj"].t()).view(S, num_heads, nope + v_dim)
    k_pe = _rope_gptj(latent[:, None, kv_lora_rank:], pos, theta)
    q_pe = _rope_gptj(q[..., nope:], pos, theta)
    qf = torch.cat([q[..., :nope], q_pe], -1)
    kf = torch.cat([kv[..., :nope], k_pe.expand(S, num_heads, rope)], -1)
    s = torch.einsum("thd,shd->hts", qf, kf) * (nope + rope) ** -0.5
    s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1)[None], float("-inf"))
    o = torch.einsum("hts,shd->thd", s.softmax(-1), kv[..., nope:]).reshape(S, num_heads * v_dim)
    return o @ W["o_proj"].t()


def mla_attention_absorbed(hidden, w, num_heads, nope, rope, v_dim, q_lora_rank, kv_lora_rank, theta=10000.0, eps=1e-6):
    """The same function computed the way decode computes it (forward_absorb_prepare/_core :1181-1335):
    w_kc folded into the query, multi-query attention over the latent rows [c_kv | k_pe] (keys 576
    wide, values = the first kv_lora_rank columns), w_vc applied to the output."""
    h = hidden.double()
    W = {k: v.double() for k, v in w.items()}
    S = h.shape[0]
    pos = torch.arange(S)
    if q_lora_rank is not None:
        a = h @ W["fused_qkv_a_proj_with_mqa"].t()
        q = _rms(a[:, :q_lora_rank], W["q_a_layernorm"], eps) @ W["q_b_proj"].t()
        latent = a[:, q_lora_rank:]
    else:
        q = h @ W["q_proj"].t()
        latent = h @ W["kv_a_proj_with_mqa"].t()
    q = q.view(S, num_heads, nope + rope)
    w_kc, w_vc = W["kv_b_proj"].unflatten(0, (-1, nope + v_dim)).split([nope, v_dim], dim=1)   # [H, nope, rank], [H, v, rank]
    q_nope_out = torch.einsum("thn,hnr->thr", q[..., :nope], w_kc)
    qa = torch.cat([q_nope_out, _rope_gptj(q[..., nope:], pos, theta)], -1)                    # [S, H, rank + rope]
    c_kv = _rms(latent[:, :kv_lora_rank], W["kv_a_layernorm"], eps)
    ka = torch.cat([c_kv, _rope_gptj(latent[:, None, kv_lora_rank:], pos, theta)[:, 0]], -1)  # [S, rank + rope]
    s = torch.einsum("thd,sd->hts", qa, ka) * (nope + rope) ** -0.5
    s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1)[None], float("-inf"))
    o = torch.einsum("hts,sr->thr", s.softmax(-1), c_kv)                                      # [S, H, rank]
    o = torch.einsum("thr,hvr->thv", o, w_vc).reshape(S, num_heads * v_dim)
    return o @ W["o_proj"].t()
