"""Full-size shapes of the configurations beyond the single-GPU headline, checked through the C ABI
with size-independent properties and float64 samples computed on the GPU:

  Llama-3-70B, one tensor-parallel rank of 8: projections N x K in {1280 x 8192 (qkv), 8192 x 1024 (o),
      7168 x 8192 (gate_up with fused SiLU-mul), 8192 x 3584 (down)}; decode attention with 8 query
      heads on ONE kv head, batch 16 / 64, context ~4224; the fused decode launch bit-identical to the
      unfused sequence at that shape.
  prefill at length: extend attention on 1 x 4096 and 4 x 1024 tokens with 32/8 and 8/1 heads, and a
      4096-token prompt prefilled as 2048 + 2048 with the second chunk reading the first through
      kv_indices (chunked prefill) -- sampled rows against float64, chunked == unchunked to one rounding.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
D = 128


@pytest.fixture(scope="module")
def ops():
    from sglang_custom_amd import ops as _ops
    from sglang_custom_amd import _lib

    _lib.load()
    return _ops


def _w(n, k, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(n, k, device=DEV, generator=g) * 0.02).bfloat16()


def _ordered_sum(part):
    acc = torch.zeros_like(part[0])
    for s in range(part.shape[0]):
        acc = acc + part[s]
    return acc


# ---- Llama-3-70B / TP = 8: hidden 8192, 8 q heads + 1 kv head per rank, intermediate 28672 / 8 --------
H70, HQ70, HKV70, INTER70 = 8192, 8, 1, 3584


@pytest.mark.parametrize("bs", [16, 64])
@pytest.mark.parametrize("n,k", [((HQ70 + 2 * HKV70) * D, H70), (H70, HQ70 * D), (H70, INTER70)])
def test_70b_shard_projections(ops, n, k, bs):
    w = _w(n, k, 11)
    wp = ops.pack_weight(w)
    x = torch.randn(bs, k, device=DEV).bfloat16()
    y = ops.linear_packed(x, wp, n)
    assert torch.equal(ops.linear_packed(x * 2, wp, n), y * 2)                # exact under power-of-two scaling
    for rows in ([0], [bs - 1], list(range(min(bs, 7))), list(range(bs - 1, -1, -1))):
        idx = torch.tensor(rows, device=DEV)
        assert torch.equal(ops.linear_packed(x[idx].contiguous(), wp, n), y[idx]), rows
    part = ops.linear_packed_partials(x, wp, n)
    assert part is not None and part.shape[0] >= 1
    assert torch.equal(_ordered_sum(part).bfloat16(), y)          # deferred split-K == in-kernel combine
    cols = torch.randint(0, n, (64,), device=DEV)
    ref = x.double() @ w[cols].double().t()
    assert (y[:, cols].double() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-3
    # the row-major entry point (other K split, other summation order): same values to one rounding
    yr = ops.linear(x, w)
    assert (yr[:, cols].double() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("bs", [16, 64])
def test_70b_shard_gate_up_with_fused_activation(ops, bs):
    w = _w(2 * INTER70, H70, 12)
    x = torch.randn(bs, H70, device=DEV).bfloat16()
    fused = ops.linear_packed_silu_mul(x, ops.pack_weight_gate_up(w), 2 * INTER70)
    plain = ops.silu_and_mul(ops.linear_packed(x, ops.pack_weight(w), 2 * INTER70))
    assert torch.equal(fused, plain)
    g, u = (x.double() @ w.double().t()).split(INTER70, dim=-1)
    ref = torch.nn.functional.silu(g) * u
    assert (fused.double() - ref).abs().max().item() <= 2 ** -6 * ref.abs().max().item() + 1e-3


def _kv_setup(bs, ctx, hq, hkv, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    lens = torch.randint(ctx - 128, ctx + 1, (bs,), device=DEV, generator=g, dtype=torch.int64)
    lens[0] = ctx
    total = int(lens.sum())
    kb = torch.randn(total + 1, hkv, D, device=DEV, generator=g).bfloat16()
    vb = torch.randn(total + 1, hkv, D, device=DEV, generator=g).bfloat16()
    perm = (torch.randperm(total, device=DEV, generator=g) + 1).int()
    r2t = torch.zeros(bs, ctx + 4, dtype=torch.int32, device=DEV)
    off = 0
    for b in range(bs):
        n = int(lens[b])
        r2t[b, :n] = perm[off:off + n]
        off += n
    q = torch.randn(bs, hq, D, device=DEV, generator=g).bfloat16()
    return q, kb, vb, r2t, lens


def _attend(ops, q, kb, vb, r2t, lens, splits):
    bs, hq = q.shape[0], q.shape[1]
    o = torch.empty(bs, hq, D, device=DEV, dtype=q.dtype)
    lg = torch.empty(bs, hq, splits, D, dtype=torch.float32, device=DEV)
    ls = torch.empty(bs, hq, splits, dtype=torch.float32, device=DEV)
    ops.decode_attention_paged(q, kb, vb, o, r2t, torch.arange(bs, device=DEV), lens, lg, ls, splits, D ** -0.5)
    return o


@pytest.mark.parametrize("bs", [16, 64])
def test_70b_shard_decode_attention(ops, bs):
    ctx = 4224
    q, kb, vb, r2t, lens = _kv_setup(bs, ctx, HQ70, HKV70, 13)
    planned = ops.decode_num_splits(bs, HKV70, ctx, 64)
    assert planned > 1                                        # bs x 1 kv head cannot fill 256 CUs without splitting
    o = _attend(ops, q, kb, vb, r2t, lens, planned)
    o1 = _attend(ops, q, kb, vb, r2t, lens, 1)
    assert (o.float() - o1.float()).abs().max().item() <= 2e-2
    # slot permutation: same tokens in other cache rows, same bits
    slots = kb.shape[0]
    sigma = torch.cat([torch.zeros(1, device=DEV, dtype=torch.long), torch.randperm(slots - 1, device=DEV) + 1])
    kb2, vb2 = torch.empty_like(kb), torch.empty_like(vb)
    kb2[sigma], vb2[sigma] = kb, vb
    r2t2 = sigma[r2t.long()].int()
    r2t2[r2t == 0] = 0
    assert torch.equal(_attend(ops, q, kb2, vb2, r2t2, lens, planned), o)
    for b in (0, bs // 2, bs - 1):
        n = int(lens[b])
        idx = r2t[b, :n].long()
        s = torch.einsum("gd,nd->gn", q[b].double(), kb[idx, 0].double()) * D ** -0.5
        ref = s.softmax(-1) @ vb[idx, 0].double()
        assert (o[b].double() - ref).abs().max().item() <= 1.6e-2
        assert (o1[b].double() - ref).abs().max().item() <= 1.6e-2


@pytest.mark.parametrize("bs", [16, 64])
def test_70b_shard_fused_decode_launch_is_bit_identical(ops, bs):
    ctx = 4224
    _, kb0, vb0, r2t, lens = _kv_setup(bs, ctx, HQ70, HKV70, 14)
    n = (HQ70 + 2 * HKV70) * D
    w = _w(n, H70, 15)
    wp = ops.pack_weight(w)
    x = torch.randn(bs, H70, device=DEV).bfloat16()
    positions = lens - 1
    loc = torch.stack([r2t[b, int(lens[b]) - 1] for b in range(bs)]).to(torch.int64)
    cache = torch.cat([torch.randn(ctx + 8, D // 2, device=DEV).cos(), torch.randn(ctx + 8, D // 2, device=DEV).sin()], 1).contiguous()
    splits = ops.decode_num_splits(bs, HKV70, ctx, 64)
    lg = torch.empty(bs, HQ70, splits, D, dtype=torch.float32, device=DEV)
    ls = torch.empty(bs, HQ70, splits, dtype=torch.float32, device=DEV)
    rpi = torch.arange(bs, device=DEV)
    kb, vb = kb0.clone(), vb0.clone()
    qkv = ops.linear_packed(x, wp, n)
    q, k, v = qkv.split([HQ70 * D, HKV70 * D, HKV70 * D], dim=-1)
    ops.rope_inplace(positions, q, k, D, cache, True, v=v, k_buffer=kb, v_buffer=vb, loc=loc)
    o_ref = torch.empty(bs, HQ70, D, device=DEV, dtype=torch.bfloat16)
    ops.decode_attention_paged(q.reshape(bs, HQ70 * D).contiguous().view(bs, HQ70, D), kb, vb, o_ref, r2t, rpi, lens, lg, ls,
                               splits, D ** -0.5)
    kb2, vb2 = kb0.clone(), vb0.clone()
    o = torch.empty_like(o_ref)
    part = ops.linear_packed_partials(x, wp, n)
    ops.decode_attention_fused_qkv(part, positions, loc, cache, kb2, vb2, o, r2t, rpi, lens, lg, ls, splits, D ** -0.5)
    torch.cuda.synchronize()
    assert torch.equal(kb, kb2) and torch.equal(vb, vb2) and torch.equal(o, o_ref)


# ---- prefill / chunked prefill at length -----------------------------------------------------------------
def _extend(ops, q, k_new, v_new, kb, vb, qo, kvp, kvi, max_ext):
    o = torch.empty_like(q)
    ops.extend_attention_fwd(q, k_new, v_new, o, kb, vb, qo, kvp, kvi, None, True, None, max_ext, D ** -0.5)
    return o


def _ref_rows(q, K, V, rows, hq, hkv):
    """float64 causal attention of query rows `rows` (absolute positions) of one sequence:
    q [S, hq, D] (only `rows` are read), K / V [S, hkv, D]."""
    g = hq // hkv
    out = []
    for p in rows:
        s = torch.einsum("hgd,nhd->hgn", q[p].double().view(hkv, g, D), K[: p + 1].double()) * D ** -0.5
        out.append(torch.einsum("hgn,nhd->hgd", s.softmax(-1), V[: p + 1].double()).reshape(hq, D))
    return torch.stack(out)


@pytest.mark.parametrize("hq,hkv", [(32, 8), (8, 1)])
@pytest.mark.parametrize("lens", [(4096,), (1024, 1024, 1000, 1024)])
def test_prefill_attention_at_length(ops, hq, hkv, lens):
    g = torch.Generator(device=DEV).manual_seed(21)
    T = sum(lens)
    q = torch.randn(T, hq, D, device=DEV, generator=g).bfloat16()
    k = torch.randn(T, hkv, D, device=DEV, generator=g).bfloat16()
    v = torch.randn(T, hkv, D, device=DEV, generator=g).bfloat16()
    qo = torch.zeros(len(lens) + 1, dtype=torch.int32, device=DEV)
    qo[1:] = torch.cumsum(torch.tensor(lens, device=DEV), 0)
    kvp = torch.zeros(len(lens) + 1, dtype=torch.int32, device=DEV)
    dummy_cache = torch.zeros(2, hkv, D, device=DEV, dtype=torch.bfloat16)
    o = _extend(ops, q, k, v, dummy_cache, dummy_cache, qo, kvp, torch.zeros(1, dtype=torch.int32, device=DEV), max(lens))
    start = 0
    for n in lens:
        rows = [0, 1, 63, 64, 65, n // 2, n - 2, n - 1]
        ref = _ref_rows(q[start: start + n], k[start: start + n], v[start: start + n], rows, hq, hkv)
        got = o[start: start + n][rows].double()
        assert (got - ref).abs().max().item() <= 1.6e-2, (hq, hkv, n)
        start += n
    # a sequence's rows do not depend on what else is in the batch
    if len(lens) > 1:
        n0 = lens[0]
        qo1 = torch.tensor([0, n0], dtype=torch.int32, device=DEV)
        o1 = _extend(ops, q[:n0].contiguous(), k[:n0].contiguous(), v[:n0].contiguous(), dummy_cache, dummy_cache, qo1,
                     kvp[:2].contiguous(), torch.zeros(1, dtype=torch.int32, device=DEV), n0)
        assert torch.equal(o1, o[:n0])


@pytest.mark.parametrize("hq,hkv", [(32, 8), (8, 1)])
def test_chunked_prefill_2048_plus_2048_equals_one_pass(ops, hq, hkv):
    """A 4096-token prompt in two chunks: chunk 2's queries read chunk 1 from scrambled cache slots
    through kv_indices (phase 1 of the kernel) and themselves from k_extend (phase 2)."""
    g = torch.Generator(device=DEV).manual_seed(22)
    S, C = 4096, 2048
    q = torch.randn(S, hq, D, device=DEV, generator=g).bfloat16()
    k = torch.randn(S, hkv, D, device=DEV, generator=g).bfloat16()
    v = torch.randn(S, hkv, D, device=DEV, generator=g).bfloat16()
    zero = torch.zeros(1, dtype=torch.int32, device=DEV)
    dummy_cache = torch.zeros(2, hkv, D, device=DEV, dtype=torch.bfloat16)
    whole = _extend(ops, q, k, v, dummy_cache, dummy_cache, torch.tensor([0, S], dtype=torch.int32, device=DEV),
                    torch.zeros(2, dtype=torch.int32, device=DEV), zero, S)
    # chunk 1 into the cache at scrambled slots
    slots = (torch.randperm(C + 100, device=DEV, generator=g)[:C] + 1)
    kb = torch.randn(C + 101, hkv, D, device=DEV, generator=g).bfloat16()
    vb = torch.randn(C + 101, hkv, D, device=DEV, generator=g).bfloat16()
    kb[slots], vb[slots] = k[:C], v[:C]
    first = _extend(ops, q[:C].contiguous(), k[:C].contiguous(), v[:C].contiguous(), dummy_cache, dummy_cache,
                    torch.tensor([0, C], dtype=torch.int32, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV), zero, C)
    second = _extend(ops, q[C:].contiguous(), k[C:].contiguous(), v[C:].contiguous(), kb, vb,
                     torch.tensor([0, S - C], dtype=torch.int32, device=DEV), torch.tensor([0, C], dtype=torch.int32, device=DEV),
                     slots.int().contiguous(), S - C)
    assert torch.equal(first, whole[:C])
    # different tile order over the same keys: equal to one output rounding (bf16 ulp at |o| < 0.5 is 2^-9..2^-8)
    assert (second.float() - whole[C:].float()).abs().max().item() <= 2 ** -7
    rows = [C, C + 1, C + 63, C + 64, 3000, S - 1]
    ref = _ref_rows(q, k, v, rows, hq, hkv)
    assert (second[[r - C for r in rows]].double() - ref).abs().max().item() <= 1.6e-2


# ---- DeepSeek-V3 MLA attention, one tensor-parallel rank of 8 (16 of 128 heads), batch 64, context ~4096 ----
def _mla_world(bs, ctx_lo, ctx_hi, seed, dt=torch.bfloat16):
    rank, rope = 512, 64
    D_ = rank + rope
    g = torch.Generator(device=DEV).manual_seed(seed)
    lens = torch.randint(ctx_lo, ctx_hi + 1, (bs,), device=DEV, generator=g, dtype=torch.int64)
    lens[0], lens[-1] = ctx_hi, ctx_lo
    total = int(lens.sum())
    cache = torch.randn(total + 1, 1, D_, device=DEV, generator=g).to(dt)
    perm = (torch.randperm(total, device=DEV, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(bs, ctx_hi, dtype=torch.int32, device=DEV)
    off = 0
    for b in range(bs):
        n = int(lens[b])
        r2t[b, :n] = perm[off: off + n]
        off += n
    rpi = torch.arange(bs, dtype=torch.int64, device=DEV)
    q = torch.randn(bs, 16, D_, device=DEV, generator=g).to(dt)
    return rank, rope, D_, lens, cache, r2t, rpi, q, g


@pytest.mark.parametrize("bs,splits", [(64, 8), (16, 16)])
def test_deepseek_shard_mla_decode_fused_rope_is_bit_identical(ops, bs, splits):
    """MLA decode at the DeepSeek-V3 TP=8 shard shape (16 heads, 512 + 64 latent row, contexts 3900 .. 4160):
    the launch that rotates q / k_pe and writes the latent row == rope + cache write + decode, bit for bit,
    outputs and cache; sampled (sequence, head) pairs against float64 attention computed on the GPU."""
    rank, rope, D_, lens, cache0, r2t, rpi, q, g = _mla_world(bs, 3900, 4160, 11)
    loc = torch.stack([r2t[b, int(lens[b]) - 1] for b in range(bs)]).to(torch.int64)
    positions = lens - 1
    k_nope = torch.randn(bs, rank, device=DEV, generator=g).bfloat16()
    lat = torch.randn(bs, 1536 + D_, device=DEV, generator=g).bfloat16()
    k_pe_view = lat[:, 1536 + rank:]
    cs = torch.cat([torch.randn(4200, rope // 2, device=DEV, generator=g).cos(),
                    torch.randn(4200, rope // 2, device=DEV, generator=g).sin()], 1).contiguous()
    scale = D_ ** -0.5
    lg = torch.empty(bs * 16 * splits * rank, dtype=torch.float32, device=DEV)
    ls = torch.empty(bs * 16 * splits, dtype=torch.float32, device=DEV)
    tickets = torch.zeros(bs * 16, dtype=torch.int32, device=DEV)
    q_ref = q.clone()
    q_pe = q_ref[..., rank:].contiguous().view(bs, -1)
    k_pe = k_pe_view.contiguous()
    ops.rope_inplace(positions, q_pe, k_pe, rope, cs, False)
    q_ref[..., rank:] = q_pe.view(bs, 16, rope)
    cache_ref = cache0.clone()
    row = torch.cat([k_nope, k_pe], -1).view(bs, 1, D_)
    ops.set_kv_buffer(cache_ref, cache_ref, loc, row, row)
    want = torch.full((bs, 16, rank), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.decode_attention_mla(q_ref, cache_ref, want, r2t, rpi, lens, lg, ls, splits, rank, scale, tickets=tickets)
    cache = cache0.clone()
    got = torch.full_like(want, float("nan"))
    ops.decode_attention_mla_fused_rope(q, k_nope, k_pe_view, positions, loc, cs, False, cache, got, r2t, rpi, lens, lg, ls,
                                        splits, rank, scale, tickets=tickets)
    torch.cuda.synchronize()
    assert torch.equal(cache.view(torch.int16), cache_ref.view(torch.int16))
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert int(tickets.abs().sum()) == 0
    for b in (0, bs // 2, bs - 1):
        kv = cache_ref[r2t[b, : int(lens[b])].long(), 0].double()
        ref = torch.softmax((q_ref[b].double() @ kv.t()) * scale, -1) @ kv[:, :rank]
        assert (got[b].double() - ref).abs().max().item() <= 2e-2 * max(ref.abs().max().item(), 1.0), b


def test_deepseek_shard_mla_decode_over_fp8_latent_cache(ops):
    """The same shape over an e4m3fn latent cache: the fp8 LDS-ring kernel with its in-launch merge == its merge
    launch, sampled sequences against float64 attention over the widened rows; the register kernel (misaligned
    split scratch) bit for bit the same kernel over a 16-bit cache holding the widened rows."""
    bs, splits = 64, 8
    rank, rope, D_, lens, cache16, r2t, rpi, q, g = _mla_world(bs, 3900, 4160, 12)
    cache8 = torch.zeros(cache16.shape, dtype=torch.float8_e4m3fn, device=DEV)
    loc = torch.arange(cache16.shape[0], dtype=torch.int64, device=DEV)
    ops.set_kv_buffer(cache8, cache8, loc, cache16, cache16)
    wide = cache8.to(torch.bfloat16)
    scale = D_ ** -0.5
    lg = torch.empty(bs * 16 * splits * rank + 1, dtype=torch.float32, device=DEV)
    ls = torch.empty(bs * 16 * splits, dtype=torch.float32, device=DEV)
    got = torch.full((bs, 16, rank), float("nan"), dtype=torch.bfloat16, device=DEV)
    same = torch.full_like(got, float("nan"))
    tickets = torch.zeros(bs * 16, dtype=torch.int32, device=DEV)
    ops.decode_attention_mla(q, cache8, got, r2t, rpi, lens, lg, ls, splits, rank, scale, tickets=tickets)
    ops.decode_attention_mla(q, cache8, same, r2t, rpi, lens, lg, ls, splits, rank, scale)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), same.view(torch.int16)) and not tickets.any()
    reg8 = torch.full_like(got, float("nan"))
    ops.decode_attention_mla(q, cache8, reg8, r2t, rpi, lens, lg[1:], ls, splits, rank, scale)   # misaligned scratch: the register kernel
    ops.decode_attention_mla(q, wide, same, r2t, rpi, lens, lg[1:], ls, splits, rank, scale)
    torch.cuda.synchronize()
    assert torch.equal(reg8.view(torch.int16), same.view(torch.int16))
    for b in (0, 31, 63):
        kv = wide[r2t[b, : int(lens[b])].long(), 0].double()
        ref = torch.softmax((q[b].double() @ kv.t()) * scale, -1) @ kv[:, :rank]
        assert (got[b].double() - ref).abs().max().item() <= 2e-2 * max(ref.abs().max().item(), 1.0), b


def test_ragged_batch_of_64_under_the_per_sequence_split_plan(ops):
    """Llama-3-70B shard attention (8 query heads on one kv head), 64 sequences of 16 .. 4160 keys under one
    8-split launch with kv_chunk_keys: every sequence bit for bit what it gets alone under the same chunk."""
    bs, S = 64, 8
    g = torch.Generator(device=DEV).manual_seed(21)
    lens = torch.randint(16, 4161, (bs,), device=DEV, generator=g, dtype=torch.int64)
    lens[0], lens[1] = 4160, 16
    total = int(lens.sum())
    kc = torch.randn(total + 1, 1, D, device=DEV, generator=g).bfloat16()
    vc = torch.randn(total + 1, 1, D, device=DEV, generator=g).bfloat16()
    perm = (torch.randperm(total, device=DEV, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(bs, 4160, dtype=torch.int32, device=DEV)
    off = 0
    for b in range(bs):
        n = int(lens[b])
        r2t[b, :n] = perm[off: off + n]
        off += n
    rpi = torch.arange(bs, dtype=torch.int64, device=DEV)
    q = torch.randn(bs, 8, D, device=DEV, generator=g).bfloat16()
    al = torch.empty(bs * 8 * S * D, dtype=torch.float32, device=DEV)
    ll = torch.empty(bs * 8 * S, dtype=torch.float32, device=DEV)
    tickets = torch.zeros(bs * 8, dtype=torch.int32, device=DEV)
    ck = torch.tensor([(4160 + S - 1) // S], dtype=torch.int32, device=DEV)
    got = torch.full((bs, 8, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.decode_attention_paged(q, kc, vc, got, r2t, rpi, lens, al, ll, S, D ** -0.5, tickets=tickets, kv_chunk_keys=ck)
    torch.cuda.synchronize()
    for b in range(0, bs, 7):
        alone = torch.full((1, 8, D), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.decode_attention_paged(q[b: b + 1], kc, vc, alone, r2t, rpi[b: b + 1], lens[b: b + 1], al, ll, S, D ** -0.5,
                                   tickets=tickets, kv_chunk_keys=ck)
        torch.cuda.synchronize()
        assert torch.equal(alone[0].view(torch.int16), got[b].view(torch.int16)), b
        kk = kc[r2t[b, : int(lens[b])].long(), 0].double()
        vv = vc[r2t[b, : int(lens[b])].long(), 0].double()
        ref = torch.softmax((q[b].double() @ kk.t()) * D ** -0.5, -1) @ vv
        assert (got[b].double() - ref).abs().max().item() <= 1.6e-2, b


@pytest.mark.parametrize("graphs", [False, True])
def test_deepseek_v3_layer_at_its_real_shapes_matches_oracle(graphs):
    """ONE dense DeepSeek-V3 layer at the real shapes (hidden 7168, all 128 heads, q_lora 1536, latent 512 + 64, MLP
    18432, YaRN) through ModelRunner in bf16: multi-head prefill on the 192-wide matrix-core tile, then decode steps
    through the one-launch latent norms, the absorbed products, RoPE + row write inside the attention launch and the
    fused MLP, eager and under a HIP graph -- teacher-forced against the oracle's stack on the CPU."""
    import cases
    from oracle.deepseek_ref import OracleDeepseek
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.server_args import ServerArgs
    from sglang_custom_amd import bench_one_batch as bob
    from sglang_custom_amd.managers.schedule_batch import Req

    cfg = hf_config_dict("deepseek-v3-dense1-vocab2k", "bfloat16")
    w = make_synthetic_llama_weights(cfg, seed=77, dtype=torch.bfloat16)
    sa = ServerArgs(dtype="bfloat16", max_total_tokens=512, max_running_requests=8, disable_cuda_graph=not graphs,
                    cuda_graph_bs=[4])
    mr = ModelRunner(ModelConfig(cfg, dtype="bfloat16"), sa, weights={k: v.cuda() for k, v in w.items()})
    assert (mr.graph_runner is not None) == graphs
    prompts = cases.e2e_prompts(cfg["vocab_size"], (5, 33, 70))
    steps = 4
    oracle = OracleDeepseek(cfg, w, torch.bfloat16, max_total_tokens=512, max_reqs=8, context_len=256)
    ref_toks, ref_logits = oracle.generate_greedy(prompts, steps)
    reqs = [Req(i, p) for i, p in enumerate(prompts)]
    nxt, logits, batch = bob.extend(reqs, mr)
    got = [logits.float().cpu()]
    for s in range(steps - 1):
        nxt, logits = bob.decode(ref_toks[:, s].cuda(), batch, mr)
        got.append(logits.float().cpu())
    for s, (a, b) in enumerate(zip(got, ref_logits)):
        d = (a - b.float()).abs()
        scale = b.float().abs().max().item()
        assert d.max().item() <= 2e-2 * max(scale, 1.0), (s, d.max().item(), scale)   # observed: 2 bf16 steps (0.06 at 7)
        top2 = b.float().topk(2, dim=-1).values
        decisive = (top2[:, 0] - top2[:, 1]) > 4e-2 * max(scale, 1.0)
        assert torch.equal(a.argmax(-1)[decisive], b.float().argmax(-1)[decisive]), s
