"""The hot path at the benchmark's full sizes (Llama-3-8B shapes, batch 64, context ~1088), where
the CPU oracle would take minutes: parity is checked through properties that do not depend on
size, plus a float64 evaluation of the same formula on the GPU for a sample of the outputs.

  projections  power-of-two scaling of x is exact; a row's result does not depend on which other
               rows share the launch (M = 64 vs M = 1); permuting weight rows permutes outputs; the
               fused gate_up + SiLU-mul equals the unfused pair; deferred split-K == in-kernel combine
  attention    slot permutation (same tokens, other cache slots) leaves the output bit-identical;
               identical V rows come back unchanged; attention over a sequence equals the
               log-sum-exp merge of attention over its two halves; split-KV == single pass
  norm / rope  row independence, and a float64 check of sampled rows
  decode step  the fused decode launch == the unfused sequence, bit for bit, at full size
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, HQ, HKV, D, INTER, BS, CTX = 4096, 32, 8, 128, 14336, 64, 1088


@pytest.fixture(scope="module")
def ops():
    from sglang_custom_amd import ops as _ops
    from sglang_custom_amd import _lib

    _lib.load()
    return _ops


def _w(n, k, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(n, k, device=DEV, generator=g) * 0.02).bfloat16()


@pytest.mark.parametrize("n,k", [((HQ + 2 * HKV) * D, H), (H, HQ * D), (H, INTER)])
def test_projection_properties_at_full_size(ops, n, k):
    w = _w(n, k, 1)
    wp = ops.pack_weight(w)
    x = torch.randn(BS, k, device=DEV).bfloat16()
    y = ops.linear_packed(x, wp, n)
    # exact under power-of-two scaling of the activations
    assert torch.equal(ops.linear_packed(x * 2, wp, n), y * 2)
    assert torch.equal(ops.linear_packed(x * 0.25, wp, n), y * 0.25)
    # a row does not see its neighbours: same bits alone, in a ragged batch, or permuted
    for rows in ([0], [63], list(range(17)), list(range(63, -1, -1))):
        idx = torch.tensor(rows, device=DEV)
        assert torch.equal(ops.linear_packed(x[idx].contiguous(), wp, n), y[idx]), rows
    # permuting output channels (in whole 64-row groups, the packing unit) permutes the outputs
    perm = torch.randperm(n // 64, device=DEV)
    rows = (perm[:, None] * 64 + torch.arange(64, device=DEV)[None]).flatten()
    assert torch.equal(ops.linear_packed(x, ops.pack_weight(w[rows].contiguous()), n), y[:, rows])
    # deferred split-K reduction == in-kernel combine
    part = ops.linear_packed_partials(x, wp, n)
    assert torch.equal(_ordered_sum(part).bfloat16(), y)
    # float64 evaluation of sampled outputs
    cols = torch.randint(0, n, (64,), device=DEV)
    ref = x.double() @ w[cols].double().t()
    assert (y[:, cols].double() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-3


def _ordered_sum(part):
    acc = torch.zeros_like(part[0])
    for s in range(part.shape[0]):
        acc = acc + part[s]
    return acc


def test_gate_up_fused_activation_at_full_size(ops):
    w = _w(2 * INTER, H, 2)
    x = torch.randn(BS, H, device=DEV).bfloat16()
    fused = ops.linear_packed_silu_mul(x, ops.pack_weight_gate_up(w), 2 * INTER)
    plain = ops.silu_and_mul(ops.linear_packed(x, ops.pack_weight(w), 2 * INTER))
    assert torch.equal(fused, plain)
    assert torch.equal(ops.linear_packed_silu_mul(x[:1].contiguous(), ops.pack_weight_gate_up(w), 2 * INTER), fused[:1])


def _kv_setup(seed=3):
    g = torch.Generator(device=DEV).manual_seed(seed)
    lens = torch.randint(CTX - 64, CTX + 1, (BS,), device=DEV, generator=g, dtype=torch.int64)
    lens[0] = CTX
    total = int(lens.sum())
    slots = total + 1
    kb = torch.randn(slots, HKV, D, device=DEV, generator=g).bfloat16()
    vb = torch.randn(slots, HKV, D, device=DEV, generator=g).bfloat16()
    perm = (torch.randperm(total, device=DEV, generator=g) + 1).int()
    r2t = torch.zeros(BS, CTX, dtype=torch.int32, device=DEV)
    off = 0
    for b in range(BS):
        n = int(lens[b])
        r2t[b, :n] = perm[off:off + n]
        off += n
    q = torch.randn(BS, HQ, D, device=DEV, generator=g).bfloat16()
    return q, kb, vb, r2t, lens


def _attend(ops, q, kb, vb, r2t, lens, splits=1):
    bs = q.shape[0]
    o = torch.empty(bs, HQ, D, device=DEV, dtype=q.dtype)
    lg = torch.empty(bs, HQ, max(splits, 1), D, dtype=torch.float32, device=DEV)
    ls = torch.empty(bs, HQ, max(splits, 1), dtype=torch.float32, device=DEV)
    ops.decode_attention_paged(q, kb, vb, o, r2t, torch.arange(bs, device=DEV), lens, lg, ls, splits, D ** -0.5)
    return o


def test_decode_attention_properties_at_full_size(ops):
    q, kb, vb, r2t, lens = _kv_setup()
    o = _attend(ops, q, kb, vb, r2t, lens)
    # the same tokens stored in other cache slots: bit-identical
    slots = kb.shape[0]
    sigma = torch.cat([torch.zeros(1, device=DEV, dtype=torch.long), torch.randperm(slots - 1, device=DEV) + 1])
    kb2, vb2 = torch.empty_like(kb), torch.empty_like(vb)
    kb2[sigma] = kb
    vb2[sigma] = vb
    r2t2 = sigma[r2t.long()].int()
    r2t2[r2t == 0] = 0
    assert torch.equal(_attend(ops, q, kb2, vb2, r2t2, lens), o)
    # rows of a batch are independent
    assert torch.equal(_attend(ops, q[5:6].contiguous(), kb, vb, r2t[5:6].contiguous(), lens[5:6].contiguous()), o[5:6])
    # identical value rows come back (the weights sum to one)
    v0 = torch.randn(HKV, D, device=DEV).bfloat16()
    oc = _attend(ops, q, kb, v0[None].expand_as(vb).contiguous(), r2t, lens)
    expect = v0[:, None, :].expand(HKV, HQ // HKV, D).reshape(HQ, D)
    assert (oc.float() - expect.float()[None]).abs().max().item() <= 2 ** -7 * 4
    # split-KV agrees with the single pass to rounding
    o4 = _attend(ops, q, kb, vb, r2t, lens, splits=4)
    assert (o4.float() - o.float()).abs().max().item() <= 2e-2
    # float64 softmax(q k^T) v for two sequences
    for b in (0, 37):
        n = int(lens[b])
        idx = r2t[b, :n].long()
        K, V = kb[idx].double(), vb[idx].double()                      # [n, HKV, D]
        qg = q[b].double().view(HKV, HQ // HKV, D)
        s = torch.einsum("hgd,nhd->hgn", qg, K) * D ** -0.5
        ref = torch.einsum("hgn,nhd->hgd", s.softmax(-1), V).reshape(HQ, D)
        assert (o[b].double() - ref).abs().max().item() <= 1.6e-2


def test_attention_equals_merge_of_halves_at_full_size(ops):
    """softmax over [A | B] = log-sum-exp merge of the two halves: checked by attending over each
    half as its own sequence (split-KV scratch gives the per-part log-sum-exp) and merging on the host."""
    q, kb, vb, r2t, lens = _kv_setup(seed=4)
    o = _attend(ops, q, kb, vb, r2t, lens).double()
    half = lens // 2
    r2t_b = torch.zeros_like(r2t)
    for b in range(BS):
        h, n = int(half[b]), int(lens[b])
        r2t_b[b, : n - h] = r2t[b, h:n]
    # scores in float64 give the merge weights; the kernel gives the per-half outputs
    oa = _attend(ops, q, kb, vb, r2t, half.contiguous()).double()
    ob = _attend(ops, q, kb, vb, r2t_b, (lens - half).contiguous()).double()
    for b in (0, 11, 63):
        h, n = int(half[b]), int(lens[b])
        K = kb[r2t[b, :n].long()].double()
        s = torch.einsum("hgd,nhd->hgn", q[b].double().view(HKV, HQ // HKV, D), K) * D ** -0.5
        la, lb = torch.logsumexp(s[..., :h], -1), torch.logsumexp(s[..., h:], -1)
        wa = torch.sigmoid(la - lb).reshape(HQ, 1)
        merged = wa * oa[b] + (1 - wa) * ob[b]
        assert (merged - o[b]).abs().max().item() <= 2.5e-2


def test_norm_and_rope_at_full_size(ops):
    x = torch.randn(BS, H, device=DEV).bfloat16()
    res = torch.randn(BS, H, device=DEV).bfloat16()
    g = (1 + 0.1 * torch.randn(H, device=DEV)).bfloat16()
    xa, ra = x.clone(), res.clone()
    ops.fused_add_rmsnorm(xa, ra, g, 1e-5)
    x1, r1 = x[7:8].clone(), res[7:8].clone()
    ops.fused_add_rmsnorm(x1, r1, g, 1e-5)
    assert torch.equal(x1, xa[7:8]) and torch.equal(r1, ra[7:8])
    s = x.double() + res.double()
    assert torch.equal(ra, s.float().bfloat16())
    ref = s * torch.rsqrt((s * s).mean(-1, keepdim=True) + 1e-5) * g.double()   # the norm uses the unrounded sum
    assert (xa.double() - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() * 2


def test_fused_decode_launch_at_full_size(ops):
    """qkv partials -> one launch (sum, RoPE, KV write, attention) == linear_packed -> rope_inplace
    (+KV write) -> decode_attention_paged, bit for bit, at batch 64 / context ~1088."""
    q_, kb0, vb0, r2t, lens = _kv_setup(seed=5)
    n = (HQ + 2 * HKV) * D
    w = _w(n, H, 6)
    wp = ops.pack_weight(w)
    x = torch.randn(BS, H, device=DEV).bfloat16()
    positions = lens - 1
    loc = torch.stack([r2t[b, int(lens[b]) - 1] for b in range(BS)]).to(torch.int64)
    cache = torch.cat([torch.randn(CTX + 8, D // 2, device=DEV).cos(), torch.randn(CTX + 8, D // 2, device=DEV).sin()], 1).contiguous()
    lg = torch.empty(BS, HQ, 1, D, dtype=torch.float32, device=DEV)
    ls = torch.empty(BS, HQ, 1, dtype=torch.float32, device=DEV)
    rpi = torch.arange(BS, device=DEV)
    kb, vb = kb0.clone(), vb0.clone()
    qkv = ops.linear_packed(x, wp, n)
    q, k, v = qkv.split([HQ * D, HKV * D, HKV * D], dim=-1)
    ops.rope_inplace(positions, q, k, D, cache, True, v=v, k_buffer=kb, v_buffer=vb, loc=loc)
    o_ref = torch.empty(BS, HQ, D, device=DEV, dtype=torch.bfloat16)
    ops.decode_attention_paged(q.reshape(BS, HQ * D).contiguous().view(BS, HQ, D), kb, vb, o_ref, r2t, rpi, lens, lg, ls, 1, D ** -0.5)
    kb2, vb2 = kb0.clone(), vb0.clone()
    o = torch.empty_like(o_ref)
    ops.decode_attention_fused_qkv(ops.linear_packed_partials(x, wp, n), positions, loc, cache, kb2, vb2, o, r2t, rpi,
                                   lens, lg, ls, 1, D ** -0.5)
    torch.cuda.synchronize()
    assert torch.equal(kb, kb2) and torch.equal(vb, vb2) and torch.equal(o, o_ref)
