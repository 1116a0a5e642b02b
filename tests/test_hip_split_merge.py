"""Split-KV decode attention with the merge inside the attention launch (arrival counters handed in as
`tickets`) against the same call without them (a merge kernel behind the attention kernel): outputs bit
for bit, every counter zero again afterwards, back-to-back calls on one counter array, and against the
CPU oracle.  Covers the matrix-core kernel (head_dim 128, 16-bit; GQA 1 / 2 / 4 / 8, both addressing
modes, an fp8 cache), the launch fused with the qkv reduction / RoPE / cache write, the MLA kernel, and
the shapes that keep the second launch (more than 64 splits, head_dim 64)."""
import pytest
import torch

import cases
from oracle import ops_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from sglang_custom_amd import ops as _ops
    from sglang_custom_amd import _lib

    _lib.load()
    return _ops


def _paged(seed, dt, B, Hq, Hkv, D, lens):
    kc, vc, r2t, pool = cases.paged_inputs(seed, dt, B, Hkv, D, lens)
    q = cases.randn((B, Hq, D), seed + 5, dt)
    return q, kc, vc, r2t, pool, torch.tensor(lens, dtype=torch.int64)


def _bits(x):
    return x.contiguous().view(torch.int16)


@pytest.mark.parametrize("dtn", ["bf16", "f16"])
def test_in_kernel_merge_equals_merge_launch(ops, dtn):
    dt = cases.DTYPES[dtn]
    D = 128
    # (heads, kv heads, lengths, splits): empty splits (length < splits), a length of 1, lengths around the
    # 16-key tile and the 64-key iteration, one kv head per rank (the TP = 8 shard), the largest merged count
    configs = [(32, 8, [700, 333, 512, 64, 129, 1], 8), (8, 1, [1100, 17, 2, 640], 3), (8, 2, [150, 47], 2),
               (4, 4, [90, 5, 300], 4), (16, 2, [65, 1000], 64), (8, 8, [33], 5)]
    tickets = torch.zeros(4096, dtype=torch.int32, device=DEV)
    for n, (Hq, Hkv, lens, splits) in enumerate(configs):
        B = len(lens)
        cpu = _paged(700 + 10 * n, dt, B, Hq, Hkv, D, lens)
        q, kc, vc, r2t, pool, lt = (x.to(DEV) for x in cpu)
        al = torch.empty((B, Hq, splits, D), dtype=torch.float32, device=DEV)
        ll = torch.empty((B, Hq, splits), dtype=torch.float32, device=DEV)
        want = torch.full((B, Hq, D), float("nan"), dtype=dt, device=DEV)
        ops.decode_attention_paged(q, kc, vc, want, r2t, pool, lt, al, ll, splits, D ** -0.5)
        for rep in range(2):   # the second call finds the counters as the first one left them
            al.fill_(float("nan"))
            ll.fill_(float("nan"))
            got = torch.full_like(want, float("nan"))
            ops.decode_attention_paged(q, kc, vc, got, r2t, pool, lt, al, ll, splits, D ** -0.5, tickets=tickets)
            torch.cuda.synchronize()
            assert not torch.isnan(got.float()).any(), (dtn, n, rep)
            assert torch.equal(_bits(got), _bits(want)), (dtn, n, rep, (got.float() - want.float()).abs().max().item())
            assert int(tickets.abs().sum()) == 0, (dtn, n, rep, "counters not handed back zeroed")
        ref = ops_ref.decode_attention(*cpu[:5], cpu[5], D ** -0.5, Hq != Hkv)
        err = (got.cpu().double() - ref.double()).abs().max().item()
        assert err <= (1.6e-2 if dtn == "bf16" else 2e-3), (dtn, n, err)


def test_in_kernel_merge_kv_indices_mode_and_fp8_cache(ops):
    dt, D, Hq, Hkv = torch.bfloat16, 128, 16, 4
    lens = [300, 1, 77, 512]
    B = len(lens)
    cpu = _paged(810, dt, B, Hq, Hkv, D, lens)
    q, kc, vc, r2t, pool, lt = (x.to(DEV) for x in cpu)
    tickets = torch.zeros(B * Hq, dtype=torch.int32, device=DEV)
    al = torch.empty((B, Hq, 8, D), dtype=torch.float32, device=DEV)
    ll = torch.empty((B, Hq, 8), dtype=torch.float32, device=DEV)
    # kv_indptr / kv_indices addressing (the Triton backend's call)
    indptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    indptr[1:] = torch.cumsum(lt, 0)
    indices = torch.cat([r2t[pool[b], : lens[b]] for b in range(B)]).contiguous()
    for splits in (2, 6):
        want = torch.full((B, Hq, D), float("nan"), dtype=dt, device=DEV)
        got = torch.full_like(want, float("nan"))
        ops.decode_attention_fwd(q, kc, vc, want, indptr, indices, al, ll, splits, 8, D ** -0.5)
        ops.decode_attention_fwd(q, kc, vc, got, indptr, indices, al, ll, splits, 8, D ** -0.5, tickets=tickets)
        torch.cuda.synchronize()
        assert torch.equal(_bits(got), _bits(want)), splits
        assert int(tickets.abs().sum()) == 0
    # e4m3 cache
    k8, v8 = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn)
    want = torch.full((B, Hq, D), float("nan"), dtype=dt, device=DEV)
    got = torch.full_like(want, float("nan"))
    ops.decode_attention_paged(q, k8, v8, want, r2t, pool, lt, al, ll, 4, D ** -0.5)
    ops.decode_attention_paged(q, k8, v8, got, r2t, pool, lt, al, ll, 4, D ** -0.5, tickets=tickets)
    torch.cuda.synchronize()
    assert not torch.isnan(got.float()).any() and torch.equal(_bits(got), _bits(want))
    assert int(tickets.abs().sum()) == 0


def test_shapes_that_keep_the_merge_launch_ignore_the_counters(ops):
    """More than 64 splits, and the head_dim-64 kernel: the counters are accepted and left alone."""
    dt = torch.bfloat16
    tickets = torch.zeros(1024, dtype=torch.int32, device=DEV)
    for (Hq, Hkv, D, lens, splits) in ((8, 2, 128, [9000, 130], 70), (8, 2, 64, [300, 40], 3)):
        B = len(lens)
        cpu = _paged(830 + D, dt, B, Hq, Hkv, D, lens)
        q, kc, vc, r2t, pool, lt = (x.to(DEV) for x in cpu)
        al = torch.empty((B, Hq, splits, D), dtype=torch.float32, device=DEV)
        ll = torch.empty((B, Hq, splits), dtype=torch.float32, device=DEV)
        want = torch.full((B, Hq, D), float("nan"), dtype=dt, device=DEV)
        got = torch.full_like(want, float("nan"))
        ops.decode_attention_paged(q, kc, vc, want, r2t, pool, lt, al, ll, splits, D ** -0.5)
        ops.decode_attention_paged(q, kc, vc, got, r2t, pool, lt, al, ll, splits, D ** -0.5, tickets=tickets)
        torch.cuda.synchronize()
        assert not torch.isnan(got.float()).any() and torch.equal(_bits(got), _bits(want)), (D, splits)
        assert int(tickets.abs().sum()) == 0


@pytest.mark.parametrize("dtype,cache_dtype", [(torch.bfloat16, torch.float32), (torch.float16, torch.float16)])
def test_fused_qkv_launch_with_in_kernel_merge(ops, dtype, cache_dtype):
    """The decode launch that starts from the qkv GEMM's partial sums, split over the keys: with counters
    == without, caches identical."""
    torch.manual_seed(5)
    D, hidden = 128, 1024
    tickets = torch.zeros(2048, dtype=torch.int32, device=DEV)
    for hq, hkv, bs, maxlen, splits in [(16, 2, 6, 530, 3), (8, 1, 7, 900, 4), (32, 8, 3, 300, 2), (8, 8, 4, 200, 8)]:
        n = (hq + 2 * hkv) * D
        wp = ops.pack_weight(torch.randn(n, hidden, device=DEV, dtype=dtype) * 0.04)
        x = torch.randn(bs, hidden, device=DEV, dtype=dtype)
        lens = torch.randint(1, maxlen, (bs,), device=DEV, dtype=torch.int64)
        lens[0] = maxlen
        lens[-1] = 2
        slots = int(lens.sum().item()) + 8
        perm = (torch.randperm(slots - 1, device=DEV) + 1).to(torch.int32)
        r2t = torch.zeros(bs + 3, maxlen + 8, dtype=torch.int32, device=DEV)
        rpi = torch.randperm(bs + 3, device=DEV)[:bs].to(torch.int64)
        off = 0
        for b in range(bs):
            L = int(lens[b])
            r2t[rpi[b], :L] = perm[off:off + L]
            off += L
        loc = torch.stack([r2t[rpi[b], int(lens[b]) - 1] for b in range(bs)]).to(torch.int64)
        positions = lens - 1
        cache = torch.cat([torch.randn(maxlen + 8, D // 2, device=DEV).cos(),
                           torch.randn(maxlen + 8, D // 2, device=DEV).sin()], dim=1).to(cache_dtype).contiguous()
        kb0 = torch.randn(slots, hkv, D, device=DEV, dtype=dtype)
        vb0 = torch.randn(slots, hkv, D, device=DEV, dtype=dtype)
        logits = torch.empty(bs, hq, splits, D, dtype=torch.float32, device=DEV)
        lse = torch.empty(bs, hq, splits, dtype=torch.float32, device=DEV)
        part = ops.linear_packed_partials(x, wp, n)
        assert part is not None
        outs = []
        for tk in (None, tickets, tickets):
            kb, vb = kb0.clone(), vb0.clone()
            o = torch.full((bs, hq, D), float("nan"), device=DEV, dtype=dtype)
            ops.decode_attention_fused_qkv(part, positions, loc, cache, kb, vb, o, r2t, rpi, lens, logits, lse, splits,
                                           D ** -0.5, tickets=tk)
            torch.cuda.synchronize()
            outs.append((o, kb, vb))
            assert int(tickets.abs().sum()) == 0
        for o, kb, vb in outs[1:]:
            assert not torch.isnan(o.float()).any()
            assert torch.equal(_bits(o), _bits(outs[0][0])), (hq, hkv, splits)
            assert torch.equal(kb, outs[0][1]) and torch.equal(vb, outs[0][2])


@pytest.mark.parametrize("dtn", ["bf16", "f16"])
def test_mla_in_kernel_merge_equals_merge_launch(ops, dtn):
    rank, rope = 512, 64
    D = rank + rope
    dt = cases.DTYPES[dtn]
    tickets = torch.zeros(4096, dtype=torch.int32, device=DEV)
    # head counts that fill, overfill and underfill the 16-head tile; empty splits; 64 splits
    for H, lens, splits in [(16, [1, 15, 300, 65], 4), (20, [129, 700], 3), (128, [1000, 3], 2), (5, [40], 8), (16, [2100], 64)]:
        total = sum(lens)
        rows = cases.randn((total + 1, 1, D), 21, dt)
        kvb = rows.to(DEV).contiguous()
        r2t = torch.zeros((len(lens), max(lens)), dtype=torch.int32)
        perm = torch.randperm(total, generator=torch.Generator().manual_seed(3)) + 1
        p = 0
        for b, n in enumerate(lens):
            r2t[b, :n] = perm[p: p + n].int()
            p += n
        rpi = torch.arange(len(lens), dtype=torch.int64).to(DEV)
        q = cases.randn((len(lens), H, D), 22, dt).to(DEV)
        lt = torch.tensor(lens, dtype=torch.int64).to(DEV)
        lg = torch.empty(len(lens) * H * splits * rank, dtype=torch.float32, device=DEV)
        ls = torch.empty(len(lens) * H * splits, dtype=torch.float32, device=DEV)
        want = torch.full((len(lens), H, rank), float("nan"), dtype=dt, device=DEV)
        ops.decode_attention_mla(q, kvb, want, r2t.to(DEV), rpi, lt, lg, ls, splits, rank, D ** -0.5)
        for rep in range(2):
            lg.fill_(float("nan"))
            ls.fill_(float("nan"))
            got = torch.full_like(want, float("nan"))
            ops.decode_attention_mla(q, kvb, got, r2t.to(DEV), rpi, lt, lg, ls, splits, rank, D ** -0.5, tickets=tickets)
            torch.cuda.synchronize()
            assert not torch.isnan(got.float()).any(), (dtn, H, splits, rep)
            assert torch.equal(_bits(got), _bits(want)), (dtn, H, splits, rep, (got.float() - want.float()).abs().max().item())
            assert int(tickets.abs().sum()) == 0


def test_undersized_split_scratch_is_refused_before_any_launch(ops):
    """Every decode-attention wrapper refuses a scratch pair one element short of rows x splits x Dv (and one
    log-sum-exp short) with Mi355Error, launches nothing (the output keeps its fill), and keeps refusing
    under `python -O` because the check is not an assert."""
    from sglang_custom_amd._lib import Mi355Error

    dt, D, Hq, Hkv, splits = torch.bfloat16, 128, 8, 2, 4
    lens = [200, 31]
    B = len(lens)
    q, kc, vc, r2t, pool, lt = (x.to(DEV) for x in _paged(900, dt, B, Hq, Hkv, D, lens))
    need_o, need_l = B * Hq * splits * D, B * Hq * splits
    full_o = torch.zeros(need_o, dtype=torch.float32, device=DEV)
    full_l = torch.zeros(need_l, dtype=torch.float32, device=DEV)
    indptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    indptr[1:] = torch.cumsum(lt, 0)
    indices = torch.cat([r2t[pool[b], : lens[b]] for b in range(B)]).contiguous()
    # fused launch inputs
    n = (Hq + 2 * Hkv) * D
    part = torch.zeros(2, B, n, dtype=torch.float32, device=DEV)
    cache = torch.zeros(256, D, dtype=torch.float32, device=DEV)
    loc = torch.stack([r2t[pool[b], lens[b] - 1] for b in range(B)]).to(torch.int64)
    # MLA inputs
    rank, rope, H = 512, 64, 16
    kvb = torch.zeros(kc.shape[0], 1, rank + rope, dtype=dt, device=DEV)
    qm = torch.zeros(B, H, rank + rope, dtype=dt, device=DEV)
    om = torch.full((B, H, rank), 7.0, dtype=dt, device=DEV)
    mla_o = torch.zeros(B * H * splits * rank, dtype=torch.float32, device=DEV)
    mla_l = torch.zeros(B * H * splits, dtype=torch.float32, device=DEV)

    o = torch.full((B, Hq, D), 7.0, dtype=dt, device=DEV)
    calls = {
        "paged": lambda a, l: ops.decode_attention_paged(q, kc, vc, o, r2t, pool, lt, a, l, splits, D ** -0.5),
        "fwd": lambda a, l: ops.decode_attention_fwd(q, kc, vc, o, indptr, indices, a, l, splits, 8, D ** -0.5),
        "fused": lambda a, l: ops.decode_attention_fused_qkv(part, lt - 1, loc, cache, kc, vc, o, r2t, pool, lt, a, l,
                                                             splits, D ** -0.5),
    }
    for name, call in calls.items():
        for a, l in ((full_o[:-1], full_l), (full_o, full_l[:-1]), (None, None), (full_o.double(), full_l.double())):
            with pytest.raises(Mi355Error):
                call(a, l)
        torch.cuda.synchronize()
        assert bool((o == 7.0).all()), f"{name}: something was launched despite the refusal"
    for a, l in ((mla_o[:-1], mla_l), (mla_o, mla_l[:-1]), (None, None)):
        with pytest.raises(Mi355Error):
            ops.decode_attention_mla(qm, kvb, om, r2t, pool, lt, a, l, splits, rank, (rank + rope) ** -0.5)
    torch.cuda.synchronize()
    assert bool((om == 7.0).all())
    # a counter array too short for rows x heads is refused the same way
    with pytest.raises(Mi355Error):
        ops.decode_attention_paged(q, kc, vc, o, r2t, pool, lt, full_o, full_l, splits, D ** -0.5,
                                   tickets=torch.zeros(B * Hq - 1, dtype=torch.int32, device=DEV))
    # the exact size is accepted
    ops.decode_attention_paged(q, kc, vc, o, r2t, pool, lt, full_o, full_l, splits, D ** -0.5)
    torch.cuda.synchronize()
    assert not bool((o == 7.0).all())


@pytest.mark.parametrize("in_kernel_merge", [False, True])
def test_per_sequence_split_plan_on_a_ragged_batch(ops, in_kernel_merge):
    """kv_chunk_keys: one int32 on the device = the batch's keys per split; sequence b runs
    ceil(len(b) / chunk) splits instead of num_kv_splits (the reference's per-sequence num_kv_splits,
    triton_backend.py:875-924).  A ragged batch of 64 sequences (16 .. 4096 keys) under one 8-split launch:
    every sequence's output is bit for bit what that sequence gets alone from a launch with as many splits as
    the plan gave it; the chunk is read when the kernel runs (changing the device value changes the plan of a
    replayed graph); a chunk too small to cover a sequence falls back to the even cut."""
    dt, D, Hq, Hkv, S = torch.bfloat16, 128, 8, 2, 8
    g = torch.Generator().manual_seed(77)
    lens = [16, 4096, 17, 700] + torch.randint(16, 4097, (60,), generator=g).tolist()
    B = len(lens)
    q, kc, vc, r2t, pool, lt = (x.to(DEV) for x in _paged(950, dt, B, Hq, Hkv, D, lens))
    al = torch.empty((B, Hq, S, D), dtype=torch.float32, device=DEV)
    ll = torch.empty((B, Hq, S), dtype=torch.float32, device=DEV)
    tickets = torch.zeros(B * Hq, dtype=torch.int32, device=DEV) if in_kernel_merge else None
    chunk = (max(lens) + S - 1) // S
    ck = torch.tensor([chunk], dtype=torch.int32, device=DEV)
    got = torch.full((B, Hq, D), float("nan"), dtype=dt, device=DEV)
    ops.decode_attention_paged(q, kc, vc, got, r2t, pool, lt, al, ll, S, D ** -0.5, tickets=tickets, kv_chunk_keys=ck)
    torch.cuda.synchronize()
    assert not torch.isnan(got.float()).any()
    # the partials of the splits a sequence does not use are empty (-inf log-sum-exp)
    for b in (0, 2, 3):
        used = (lens[b] + chunk - 1) // chunk
        assert torch.isinf(ll[b, :, used:]).all() and torch.isfinite(ll[b, :, :used]).all(), b
    # per request, alone, with the split count the plan gave it and an even cut of ITS length when that equals the chunk cut
    one_al = torch.empty((1, Hq, S, D), dtype=torch.float32, device=DEV)
    one_ll = torch.empty((1, Hq, S), dtype=torch.float32, device=DEV)
    for b in range(B):
        alone = torch.full((1, Hq, D), float("nan"), dtype=dt, device=DEV)
        ops.decode_attention_paged(q[b: b + 1], kc, vc, alone, r2t, pool[b: b + 1], lt[b: b + 1], one_al, one_ll, S, D ** -0.5,
                                   kv_chunk_keys=ck)
        torch.cuda.synchronize()
        assert torch.equal(_bits(alone[0]), _bits(got[b])), (b, lens[b])
    # single-split sequences equal the unsplit kernel bit for bit only up to the merge's rounding: check against the oracle instead
    cpu = [x.cpu() for x in (q, kc, vc, r2t, pool)]
    ref = ops_ref.decode_attention(*cpu, torch.tensor(lens, dtype=torch.int64), D ** -0.5, True)
    assert (got.cpu().double() - ref.double()).abs().max().item() <= 1.6e-2
    # the value is read at run time: a graph captured with one chunk follows the device value on replay
    out = torch.full_like(got, float("nan"))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.decode_attention_paged(q, kc, vc, out, r2t, pool, lt, al, ll, S, D ** -0.5, tickets=tickets, kv_chunk_keys=ck)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            ops.decode_attention_paged(q, kc, vc, out, r2t, pool, lt, al, ll, S, D ** -0.5, tickets=tickets, kv_chunk_keys=ck)
    torch.cuda.current_stream().wait_stream(s)
    ck.fill_(1)          # too small for any sequence: every sequence falls back to the even 8-way cut
    gr.replay()
    torch.cuda.synchronize()
    even = torch.full_like(got, float("nan"))
    ops.decode_attention_paged(q, kc, vc, even, r2t, pool, lt, al, ll, S, D ** -0.5, tickets=tickets)
    torch.cuda.synchronize()
    assert torch.equal(_bits(out), _bits(even))
    ck.fill_(chunk)
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(_bits(out), _bits(got))
    if tickets is not None:
        assert int(tickets.abs().sum()) == 0
