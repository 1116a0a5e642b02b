"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle on the seeded
inputs of tests/golden/cases.py, and against the reference-generated golden vectors.
Integer / index work must be bit-exact; floating point within the tolerance written at each
test (fp32 paths 1e-5-level; bf16/f16 one-rounding-level, as the reference's own kernel tests:
sgl-kernel/tests/test_norm.py 1e-3 fp16, test_rotary_embedding.py 1e-2 bf16,
test/srt/cpu/test_decode.py 3e-2)."""
import math

import numpy as np
import pytest
import torch

import cases
from oracle import ops_ref

pytestmark = pytest.mark.gpu
DT = cases.DTYPES
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from sglang_custom_amd import ops as _ops
    from sglang_custom_amd import _lib

    _lib.load()
    return _ops


def t(a):
    return torch.from_numpy(np.asarray(a))


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


# one rounding step of the dtype at |x| ~ 4 (bf16 ulp 2^-5 at [4,8)), 0 for fp32 compare-in-fp32
ROUND_TOL = {"f32": 2e-6, "bf16": 3.2e-2, "f16": 4e-3}


@pytest.mark.parametrize("dtn", list(DT))
def test_rmsnorm(ops, golden_ops, dtn):
    dt = DT[dtn]
    for i in range(len(cases.NORM_CASES)):
        x, r, w = cases.norm_inputs(i, dt)
        ref = ops_ref.rmsnorm(x, w, 1e-5)
        got = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-5)
        assert maxdiff(got, ref) <= ROUND_TOL[dtn]
        assert maxdiff(got, t(golden_ops[f"rmsnorm/{dtn}/{i}"])) <= ROUND_TOL[dtn]
        # at most a handful of elements may differ by one rounding (different summation order)
        if dtn != "f32":
            assert (got.cpu() != ref).float().mean() < 0.02
        yref, rref = ops_ref.rmsnorm(x, w, 1e-5, r)
        xg, rg = x.to(DEV), r.to(DEV)
        ops.fused_add_rmsnorm(xg, rg, w.to(DEV), 1e-5)
        assert torch.equal(rg.cpu(), rref), "residual = T(x + residual) must be bit-exact"
        assert maxdiff(xg, yref) <= ROUND_TOL[dtn]
        if dtn != "f32":
            assert (xg.cpu() != yref).float().mean() < 0.02


def test_rmsnorm_strided_and_shapes(ops):
    for rows, h in ((1, 8), (19, 3072), (3, 16384), (2, 20000)):
        x = cases.randn((rows, 2 * h), 1, torch.bfloat16)[:, :h]   # row stride 2h
        w = cases.randn((h,), 2, torch.bfloat16)
        got = ops.rmsnorm(cases.randn((rows, 2 * h), 1, torch.bfloat16).to(DEV)[:, :h], w.to(DEV), 1e-6)
        assert maxdiff(got, ops_ref.rmsnorm(x, w, 1e-6)) <= 6.3e-2
    assert ops.rmsnorm(torch.empty(0, 64, device=DEV), torch.ones(64, device=DEV), 1e-6).shape == (0, 64)


@pytest.mark.parametrize("dtn", list(DT))
def test_silu_and_mul(ops, golden_ops, dtn):
    for i in range(len(cases.ACT_CASES)):
        x = cases.act_inputs(i, DT[dtn])
        got = ops.silu_and_mul(x.to(DEV))
        ref = ops_ref.silu_and_mul(x)
        assert maxdiff(got, ref) <= ROUND_TOL[dtn]
        assert maxdiff(got, t(golden_ops[f"silu_and_mul/{dtn}/{i}"])) <= ROUND_TOL[dtn]
        if dtn != "f32":
            assert (got.cpu() != ref).float().mean() < 0.02   # expf vs the CPU's vectorised exp


@pytest.mark.parametrize("dtn", list(DT))
def test_rope_and_kv_write(ops, golden_ops, dtn):
    dt = DT[dtn]
    for i, (T, hq, hkv, hd, rd, neox, l3) in enumerate(cases.ROPE_CASES):
        pos, q, k = cases.rope_inputs(i, dt)
        cache = ops_ref.cos_sin_cache(rd, 2048, 500000.0 if l3 else 10000.0, cases.LLAMA3 if l3 else None, dt)
        qref, kref = ops_ref.apply_rope(pos, q, k, hd, cache, neox)
        qg, kg = q.to(DEV), k.to(DEV)
        ops.rope_inplace(pos.to(DEV), qg, kg, hd, cache.to(DEV), neox)
        # same products, same roundings as the reference's native formula: bit-exact
        assert torch.equal(qg.cpu(), qref) and torch.equal(kg.cpu(), kref)
        assert torch.equal(qg.cpu().float(), t(golden_ops[f"rope/{dtn}/{i}/q"]))
        # fused KV write, on a strided qkv like the model produces
        qkv = torch.cat([q, k, cases.randn((T, hkv * hd), 530 + i, dt)], dim=1).to(DEV)
        q2, k2, v2 = qkv.split([hq * hd, hkv * hd, hkv * hd], dim=-1)
        kb = torch.zeros((40, hkv, hd), dtype=dt, device=DEV)
        vb = torch.zeros_like(kb)
        loc = (torch.randperm(39)[:T] + 1).to(DEV)
        ops.rope_inplace(pos.to(DEV), q2, k2, hd, cache.to(DEV), neox, v=v2, k_buffer=kb, v_buffer=vb, loc=loc)
        assert torch.equal(q2.cpu(), qref) and torch.equal(k2.cpu(), kref)
        assert torch.equal(kb[loc].view(T, -1).cpu(), kref) and torch.equal(vb[loc].view(T, -1), v2)
        assert kb[0].abs().sum() == 0  # sink row untouched


def test_set_kv_buffer_and_embedding(ops):
    for dt in DT.values():
        k = cases.randn((11, 2, 64), 1, dt).to(DEV)
        v = cases.randn((11, 2, 64), 2, dt).to(DEV)
        kb = torch.zeros((30, 2, 64), dtype=dt, device=DEV)
        vb = torch.zeros_like(kb)
        for loc in ((torch.randperm(29)[:11] + 1).to(DEV), (torch.randperm(29)[:11] + 1).to(DEV).int()):
            ops.set_kv_buffer(kb, vb, loc, k, v)
            assert torch.equal(kb[loc.long()], k) and torch.equal(vb[loc.long()], v)
        tab = cases.randn((100, 48), 3, dt).to(DEV)
        ids = torch.randint(0, 100, (17,)).to(DEV)
        assert torch.equal(ops.embedding(ids, tab), tab[ids])
        part = ops.embedding(ids, tab[25:50].contiguous(), 25, 50)
        ref = torch.where(((ids >= 25) & (ids < 50))[:, None], tab[ids], torch.zeros_like(tab[ids]))
        assert torch.equal(part, ref)


def test_index_ops_exact(ops):
    g = torch.Generator().manual_seed(5)
    for bs in (1, 37, 300):
        r2t = torch.randint(1, 10 ** 6, (bs + 5, 600), generator=g, dtype=torch.int32)
        pool = torch.randperm(bs + 5, generator=g)[:bs]
        lens = torch.randint(1, 600, (bs,), generator=g)
        indptr = torch.zeros(bs + 1, dtype=torch.int32)
        indptr[1:] = torch.cumsum(lens, 0)
        start = torch.randint(0, 3, (bs,), generator=g, dtype=torch.int32)
        lens2 = torch.clamp(lens - 3, min=0)
        for ptype in (torch.int64, torch.int32):
            out = torch.empty(int(lens.sum()), dtype=torch.int32, device=DEV)
            ops.create_kv_indices(r2t.to(DEV), pool.to(ptype).to(DEV), lens.to(DEV), indptr.to(DEV), None, out)
            assert torch.equal(out.cpu(), ops_ref.create_kv_indices(r2t, pool, lens))
        indptr2 = torch.zeros(bs + 1, dtype=torch.int32)
        indptr2[1:] = torch.cumsum(lens2, 0)
        out = torch.empty(max(int(lens2.sum()), 1), dtype=torch.int32, device=DEV)
        ops.create_kv_indices(r2t.to(DEV), pool.to(DEV), lens2.int().to(DEV), indptr2.to(DEV), start.to(DEV), out)
        assert torch.equal(out.cpu()[: int(lens2.sum())], ops_ref.create_kv_indices(r2t, pool, lens2, start))
        pre = torch.randint(0, 100, (bs,), generator=g)
        ext = torch.randint(1, 400, (bs,), generator=g)
        seq = pre + ext
        ocl = torch.randint(1, 10 ** 6, (int(ext.sum()),), generator=g)
        a = r2t.clone().to(DEV)
        ops.write_req_to_token(a, pool.to(DEV), pre.to(DEV), seq.to(DEV), ext.to(DEV), ocl.to(DEV))
        b = r2t.clone()
        ops_ref.write_req_to_token(b, pool, pre, seq, ocl)
        assert torch.equal(a.cpu(), b)
        p, s = ops.compute_position(pre.int().to(DEV), ext.int().to(DEV), int(ext.sum()))
        pr, sr = ops_ref.compute_position(pre.int(), ext.int())
        assert torch.equal(p.cpu(), pr) and torch.equal(s.cpu(), sr)
        sl = seq.clone().to(DEV)
        pos = torch.empty(bs, dtype=torch.int64, device=DEV)
        oc = torch.randint(1, 10 ** 6, (bs,), generator=g)
        ops.prepare_decode(a, pool.to(DEV), sl, oc.to(DEV), pos)
        b[pool, seq] = oc.int()
        assert torch.equal(a.cpu(), b) and torch.equal(sl.cpu(), seq + 1) and torch.equal(pos.cpu(), ops_ref.clamp_position(seq + 1))


def _decode_run(ops, q, kc, vc, r2t, pool, lens, splits, cap=0.0, mode="paged"):
    B, Hq, D = q.shape
    o = torch.empty((B, Hq, vc.shape[-1]), dtype=q.dtype, device=DEV)
    al = torch.empty((B, Hq, splits, vc.shape[-1]), dtype=torch.float32, device=DEV)
    ll = torch.empty((B, Hq, splits), dtype=torch.float32, device=DEV)
    scale = D ** -0.5
    if mode == "paged":
        ops.decode_attention_paged(q.to(DEV), kc.to(DEV), vc.to(DEV), o, r2t.to(DEV), pool.to(DEV), lens.to(DEV), al, ll, splits, scale, cap)
    else:
        indptr = torch.zeros(B + 1, dtype=torch.int32)
        indptr[1:] = torch.cumsum(lens, 0)
        kvi = ops_ref.create_kv_indices(r2t, pool, lens).to(DEV)
        ops.decode_attention_fwd(q.to(DEV), kc.to(DEV), vc.to(DEV), o, indptr.to(DEV), kvi, al, ll, splits, splits, scale, cap)
    return o


# bf16: one rounding step of an output in [2, 4) is 2^-6
ATTN_TOL = {"f32": 2e-5, "bf16": 1.6e-2, "f16": 2e-3}


@pytest.mark.parametrize("dtn", list(DT))
def test_decode_attention(ops, golden_ops, dtn):
    for i, (B, Hq, Hkv, D, _) in enumerate(cases.DECODE_CASES):
        q, kc, vc, r2t, pool, lens = cases.decode_inputs(i, DT[dtn])
        ref = ops_ref.decode_attention(q, kc, vc, r2t, pool, lens, D ** -0.5, Hq != Hkv)
        for splits in (1, 3, 8):
            for mode in ("paged", "indices"):
                o = _decode_run(ops, q, kc, vc, r2t, pool, lens, splits, mode=mode)
                assert maxdiff(o, ref) <= ATTN_TOL[dtn], (i, splits, mode)
                assert maxdiff(o, t(golden_ops[f"decode/{dtn}/{i}"])) <= ATTN_TOL[dtn]


def test_decode_attention_shapes_and_logit_cap(ops):
    # reference test shapes (test/srt/test_triton_attention_kernels.py:293-388, cpu/test_decode.py:62-163)
    for (B, Hq, Hkv, D, S, cap) in ((2, 12, 4, 80, 100, 0.0), (3, 40, 8, 128, 500, 0.0), (2, 16, 16, 96, 128, 0.0),
                                    (2, 8, 2, 64, 77, 30.0), (1, 64, 8, 128, 1024, 0.0), (2, 6, 2, 256, 65, 0.0),
                                    (2, 4, 1, 13, 9, 0.0)):
        for dtn in ("f32", "bf16"):
            dt = DT[dtn]
            lens = [S, max(1, S // 3)][:B] + [5] * (B - 2)
            kc, vc, r2t, pool = cases.paged_inputs(42, dt, B, Hkv, D, lens)
            q = cases.randn((B, Hq, D), 43, dt)
            lt = torch.tensor(lens, dtype=torch.int64)
            for splits in (1, 4):
                o = _decode_run(ops, q, kc, vc, r2t, pool, lt, splits, cap)
                for b in range(B):
                    toks = r2t[pool[b], : lens[b]].long()
                    ref = ops_ref.attention_fp64(q[b][:, None], kc[toks].transpose(0, 1), vc[toks].transpose(0, 1), D ** -0.5, None, cap)[:, 0]
                    assert maxdiff(o[b], ref) <= ATTN_TOL[dtn], (B, Hq, Hkv, D, S, cap, splits)


def test_mla_compressed_kv_decode(ops):
    """DeepSeek-style MQA over the compressed cache (deepseek_v2.py:866-875): 16 local heads, latent
    row 512 + 64, value = first 512 features of the same row; pool writes go through
    MLATokenToKVPool; checked against the float64 softmax(q k^T) v of the oracle."""
    from sglang_custom_amd.mem_cache.memory_pool import MLATokenToKVPool

    class _L:
        layer_id = 0

    H, rank, rope = 16, 512, 64
    D = rank + rope
    for dtn in ("bf16", "f32"):
        dt = DT[dtn]
        lens = [200, 1, 77]
        pool = MLATokenToKVPool(400, 1, dt, rank, rope, 1, DEV)
        g = torch.Generator().manual_seed(11)
        slots = (torch.randperm(399, generator=g)[: sum(lens)] + 1)
        rows = cases.randn((sum(lens), 1, D), 12, dt)
        half = sum(lens) // 2
        pool.set_kv_buffer(_L, slots[:half].to(DEV), rows[:half].to(DEV))
        pool.set_mla_kv_buffer(_L, slots[half:].to(DEV), rows[half:, :, :rank].to(DEV), rows[half:, :, rank:].to(DEV))
        assert torch.equal(pool.get_key_buffer(0)[slots.to(DEV)].cpu(), rows)
        assert pool.get_value_buffer(0).shape[-1] == rank and pool.get_key_buffer(0)[0].abs().sum() == 0
        r2t = torch.zeros((4, 256), dtype=torch.int32)
        p = 0
        for b, n in enumerate(lens):
            r2t[b + 1, :n] = slots[p: p + n].int()
            p += n
        rpi = torch.tensor([1, 2, 3], dtype=torch.int64)
        q = cases.randn((3, H, D), 13, dt)
        for splits in (1, 3):
            o = torch.empty((3, H, rank), dtype=dt, device=DEV)
            lg = torch.empty(3 * H * splits * rank, dtype=torch.float32, device=DEV)
            ls = torch.empty(3 * H * splits, dtype=torch.float32, device=DEV)
            ops.decode_attention_mla(q.to(DEV), pool.get_key_buffer(0), o, r2t.to(DEV), rpi.to(DEV),
                                     torch.tensor(lens, dtype=torch.int64).to(DEV), lg, ls, splits, rank, D ** -0.5)
            p = 0
            for b, n in enumerate(lens):
                kv = rows[p: p + n, 0]
                ref = ops_ref.attention_fp64(q[b][:, None], kv[None], kv[None, :, :rank], D ** -0.5, None)[:, 0]
                assert maxdiff(o[b], ref) <= ATTN_TOL[dtn], (dtn, b, splits)
                p += n


def _extend_run(ops, q, kc, vc, r2t, pool, seq, pre, ext, Hkv, D):
    """New-token K/V are read from the cache rows (they are already there) as k_extend/v_extend."""
    T = q.shape[0]
    kext = torch.empty((T, Hkv, D), dtype=q.dtype)
    vext = torch.empty((T, Hkv, D), dtype=q.dtype)
    p = 0
    for b in range(len(seq)):
        toks = r2t[pool[b], int(pre[b]): int(seq[b])].long()
        kext[p: p + len(toks)] = kc[toks]
        vext[p: p + len(toks)] = vc[toks]
        p += len(toks)
    qo = torch.zeros(len(seq) + 1, dtype=torch.int32)
    qo[1:] = torch.cumsum(ext, 0)
    kvp = torch.zeros(len(seq) + 1, dtype=torch.int32)
    kvp[1:] = torch.cumsum(pre, 0)
    kvi = ops_ref.create_kv_indices(r2t, pool, pre) if int(pre.sum()) else torch.zeros(1, dtype=torch.int32)
    o = torch.empty((T, q.shape[1], D), dtype=q.dtype, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), kext.to(DEV), vext.to(DEV), o, kc.to(DEV), vc.to(DEV), qo.to(DEV),
                             kvp.to(DEV), kvi.to(DEV), None, True, None, int(ext.max()), D ** -0.5)
    return o


@pytest.mark.parametrize("dtn", list(DT))
def test_extend_attention(ops, golden_ops, dtn):
    for i, (Hq, Hkv, D, pre, ext) in enumerate(cases.EXTEND_CASES):
        q, kc, vc, r2t, pool, seq, pre_t, ext_t = cases.extend_inputs(i, DT[dtn])
        ref = ops_ref.extend_attention(q, kc, vc, r2t, pool, seq, pre_t, ext_t, D ** -0.5, Hq != Hkv, True)
        o = _extend_run(ops, q, kc, vc, r2t, pool, seq, pre_t, ext_t, Hkv, D)
        assert maxdiff(o, ref) <= ATTN_TOL[dtn], i
        assert maxdiff(o, t(golden_ops[f"extend/{dtn}/{i}"])) <= ATTN_TOL[dtn]


def test_extend_attention_reference_test_shapes(ops):
    # test/srt/test_triton_attention_kernels.py:36-181: Hq=12, Hkv=4, D in {128, 96, 80, 13}
    for D in (128, 96, 80, 13):
        pre, ext = [31, 0, 200], [65, 130, 7]
        seq = [a + b for a, b in zip(pre, ext)]
        for dtn in ("f32", "bf16"):
            dt = DT[dtn]
            kc, vc, r2t, pool = cases.paged_inputs(77, dt, 3, 4, D, seq)
            q = cases.randn((sum(ext), 12, D), 78, dt)
            o = _extend_run(ops, q, kc, vc, r2t, pool, torch.tensor(seq), torch.tensor(pre), torch.tensor(ext), 4, D)
            p = 0
            for b in range(3):
                toks = r2t[pool[b], : seq[b]].long()
                ref = ops_ref.attention_fp64(q[p: p + ext[b]].transpose(0, 1), kc[toks].transpose(0, 1),
                                             vc[toks].transpose(0, 1), D ** -0.5, pre[b]).transpose(0, 1)
                assert maxdiff(o[p: p + ext[b]], ref) <= ATTN_TOL[dtn], (D, b)
                p += ext[b]


@pytest.mark.parametrize("dtn", ["bf16", "f16"])
def test_extend_attention_matrix_core_tile(ops, dtn):
    """head_dim 128, 16-bit goes through the MFMA flash-attention tile: ragged lengths around the
    64-row / 64-key tile edges, long prefixes with extend length 1 (the MIXED-batch row), MHA and
    GQA, causal and full attention."""
    dt = DT[dtn]
    for (Hq, Hkv), (pre, ext) in [((8, 2), ([0, 63, 64, 300], [64, 65, 1, 129])),
                                  ((4, 4), ([129, 0], [1, 257])),
                                  ((16, 2), ([17], [200]))]:
        seq = [a + b for a, b in zip(pre, ext)]
        kc, vc, r2t, pool = cases.paged_inputs(55, dt, len(seq), Hkv, 128, seq)
        q = cases.randn((sum(ext), Hq, 128), 56, dt)
        o = _extend_run(ops, q, kc, vc, r2t, pool, torch.tensor(seq), torch.tensor(pre), torch.tensor(ext), Hkv, 128)
        p = 0
        for b in range(len(seq)):
            toks = r2t[pool[b], : seq[b]].long()
            ref = ops_ref.attention_fp64(q[p: p + ext[b]].transpose(0, 1), kc[toks].transpose(0, 1),
                                         vc[toks].transpose(0, 1), 128 ** -0.5, pre[b]).transpose(0, 1)
            assert maxdiff(o[p: p + ext[b]], ref) <= ATTN_TOL[dtn], (Hq, Hkv, b)
            p += ext[b]
    # full (non-causal) attention over prefix + extend
    Hq, Hkv, pre, ext = 8, 2, [40, 0], [70, 33]
    seq = [a + b for a, b in zip(pre, ext)]
    kc, vc, r2t, pool = cases.paged_inputs(57, dt, 2, Hkv, 128, seq)
    q = cases.randn((sum(ext), Hq, 128), 58, dt)
    kext = torch.cat([kc[r2t[pool[b], pre[b]: seq[b]].long()] for b in range(2)])
    vext = torch.cat([vc[r2t[pool[b], pre[b]: seq[b]].long()] for b in range(2)])
    qo = torch.tensor([0, ext[0], sum(ext)], dtype=torch.int32)
    kvp = torch.tensor([0, pre[0], sum(pre)], dtype=torch.int32)
    kvi = ops_ref.create_kv_indices(r2t, pool, torch.tensor(pre))
    o = torch.empty((sum(ext), Hq, 128), dtype=dt, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), kext.to(DEV), vext.to(DEV), o, kc.to(DEV), vc.to(DEV), qo.to(DEV), kvp.to(DEV),
                             kvi.to(DEV), None, False, None, max(ext), 128 ** -0.5)
    p = 0
    for b in range(2):
        toks = r2t[pool[b], : seq[b]].long()
        ref = ops_ref.attention_fp64(q[p: p + ext[b]].transpose(0, 1), kc[toks].transpose(0, 1),
                                     vc[toks].transpose(0, 1), 128 ** -0.5, None).transpose(0, 1)
        assert maxdiff(o[p: p + ext[b]], ref) <= ATTN_TOL[dtn], b
        p += ext[b]


def test_extend_attention_custom_mask_equals_causal(ops):
    # the reference checks causal flag == explicit lower-triangular custom mask (same file :120-181)
    Hq, Hkv, D = 8, 2, 64
    pre, ext = [5, 0], [9, 33]
    seq = [a + b for a, b in zip(pre, ext)]
    dt = torch.bfloat16
    kc, vc, r2t, pool = cases.paged_inputs(91, dt, 2, Hkv, D, seq)
    q = cases.randn((sum(ext), Hq, D), 92, dt)
    base = _extend_run(ops, q, kc, vc, r2t, pool, torch.tensor(seq), torch.tensor(pre), torch.tensor(ext), Hkv, D)
    masks, indptr = [], [0]
    for p_, e_ in zip(pre, ext):
        m = torch.zeros((e_, p_ + e_), dtype=torch.bool)
        m[:, :p_] = True
        m[:, p_:] = torch.tril(torch.ones(e_, e_, dtype=torch.bool))
        masks.append(m.flatten())
        indptr.append(indptr[-1] + m.numel())
    T = sum(ext)
    kext = torch.cat([kc[r2t[pool[b], pre[b]: seq[b]].long()] for b in range(2)])
    vext = torch.cat([vc[r2t[pool[b], pre[b]: seq[b]].long()] for b in range(2)])
    qo = torch.tensor([0, ext[0], T], dtype=torch.int32)
    kvp = torch.tensor([0, pre[0], pre[0] + pre[1]], dtype=torch.int32)
    kvi = ops_ref.create_kv_indices(r2t, pool, torch.tensor(pre))
    o = torch.empty((T, Hq, D), dtype=dt, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), kext.to(DEV), vext.to(DEV), o, kc.to(DEV), vc.to(DEV), qo.to(DEV), kvp.to(DEV),
                             kvi.to(DEV), torch.cat(masks).to(DEV), False, torch.tensor(indptr, dtype=torch.int64).to(DEV),
                             max(ext), D ** -0.5, skip_prefix_custom_mask=False)
    assert torch.equal(o, base)


@pytest.mark.parametrize("dtn", list(DT))
def test_linear(ops, dtn):
    dt = DT[dtn]
    shapes = [(1, 512, 512), (7, 1000, 256), (64, 768, 512), (33, 2048, 1536), (64, 1024, 4096), (100, 512, 300),
              (5, 64, 8), (16, 4100, 1024)]
    for (M, N, K) in shapes:
        x = cases.randn((M, K), 7, dt)
        w = cases.randn((N, K), 8, dt, 1.0 / math.sqrt(K))
        ref = torch.nn.functional.linear(x, w)
        exact = x.double() @ w.double().t()
        got = ops.linear(x.to(DEV), w.to(DEV))
        if dtn == "f32":
            assert maxdiff(got, exact) <= 2e-5
        else:
            # fp32-accumulated, rounded once: within one rounding of the exact product, and as close
            # to it as the CPU's own F.linear is
            assert maxdiff(got, exact) <= max(1.5 * maxdiff(ref, exact), ROUND_TOL[dtn] / 2)
    # strided input (a column slice) and fp32-widened output
    x = cases.randn((9, 1024), 9, torch.bfloat16)
    w = cases.randn((300, 512), 10, torch.bfloat16, 0.05)
    got = ops.linear(x.to(DEV)[:, :512], w.to(DEV), out_dtype=torch.float32)
    ref = (x[:, :512].double() @ w.double().t())
    assert got.dtype == torch.float32 and maxdiff(got, ref) <= 1.6e-2
    assert torch.equal(got, got.bfloat16().float()), "fp32 output must hold bf16-rounded values"


@pytest.mark.parametrize("dtn", ["bf16", "f16"])
def test_linear_packed_and_split_k(ops, dtn):
    """The packed-weight decode GEMM against the exact product, across the shapes that pick each
    split-K plan (none / in-launch combine of 2..16 partials), ragged N, every M tile count, bias,
    fp32 output; repeated calls on one workspace must give identical bits (the arrival counters
    come back zeroed) and a second, smaller call right after a larger one must not see its data."""
    dt = DT[dtn]
    shapes = [(64, 4096, 4096), (64, 1024, 14336), (33, 6144, 2048), (17, 28672 // 4, 1024), (1, 512, 512),
              (64, 1000, 256), (48, 200, 7168), (5, 70000, 256), (16, 4100, 1024), (64, 64, 32)]
    prev = None
    for i, (M, N, K) in enumerate(shapes):
        x = cases.randn((M, K), 70 + i, dt)
        w = cases.randn((N, K), 90 + i, dt, 1.0 / math.sqrt(K))
        b = cases.randn((N,), 110 + i, dt) if i % 2 else None
        exact = x.double() @ w.double().t() + (b.double() if b is not None else 0.0)
        ref = torch.nn.functional.linear(x.float(), w.float(), b.float() if b is not None else None).to(dt)
        wd = w.to(DEV)
        wp = ops.pack_weight(wd)
        assert wp.numel() == (N + 63) // 64 * 64 * K
        bd = b.to(DEV) if b is not None else None
        got = ops.linear_packed(x.to(DEV), wp, N, bd)
        assert got.shape == (M, N) and got.dtype == dt
        assert maxdiff(got, exact) <= max(1.5 * maxdiff(ref, exact), ROUND_TOL[dtn] / 2), (M, N, K)
        # the row-major kernel may split K differently (other summation order): same value to one
        # rounding; repeats of the same call give the same bits
        assert maxdiff(got, ops.linear(x.to(DEV), wd, bd)) <= ROUND_TOL[dtn], (M, N, K)
        for _ in range(3):
            assert torch.equal(got, ops.linear_packed(x.to(DEV), wp, N, bd))
        if prev is not None:
            px, pwp, pN, pb, pgot = prev
            assert torch.equal(pgot, ops.linear_packed(px, pwp, pN, pb))
        prev = (x.to(DEV), wp, N, bd, got)
        got32 = ops.linear_packed(x.to(DEV), wp, N, bd, out_dtype=torch.float32)
        assert got32.dtype == torch.float32 and torch.equal(got32, got.float())
    # strided activations (a column slice of a wider buffer)
    x = cases.randn((9, 1024), 9, dt).to(DEV)
    w = cases.randn((300, 512), 10, dt, 0.05).to(DEV)
    assert maxdiff(ops.linear_packed(x[:, :512], ops.pack_weight(w), 300), ops.linear(x[:, :512].contiguous(), w)) <= ROUND_TOL[dtn]


@pytest.mark.parametrize("dtn", ["bf16", "f16"])
def test_gate_up_projection_with_fused_silu_mul(ops, dtn):
    """SiLU-and-mul in the GEMM epilogue must give the bits of GEMM -> silu_and_mul (same
    accumulation order, same rounding points), for every M tile count and for split and
    unsplit plans; and match the oracle's silu_and_mul of the exact product to one rounding."""
    dt = DT[dtn]
    for i, (M, d, K) in enumerate([(64, 14336, 4096), (33, 1024, 512), (1, 256, 256), (17, 3584, 1024), (64, 72, 768)]):
        x = cases.randn((M, K), 300 + i, dt).to(DEV)
        w = cases.randn((2 * d, K), 320 + i, dt, 1.0 / math.sqrt(K)).to(DEV)
        fused = ops.linear_packed_silu_mul(x, ops.pack_weight_gate_up(w), 2 * d)
        unfused = ops.silu_and_mul(ops.linear_packed(x, ops.pack_weight(w), 2 * d))
        assert fused.shape == (M, d)
        assert torch.equal(fused, unfused), (M, d, K)
        exact = x.double().cpu() @ w.double().cpu().t()
        ref = ops_ref.silu_and_mul(exact.to(dt))
        assert maxdiff(fused, ref) <= 2 * ROUND_TOL[dtn]
        assert torch.equal(fused, ops.linear_packed_silu_mul(x, ops.pack_weight_gate_up(w), 2 * d))


def test_fp8_quant_and_gemm(ops, golden_ops):
    for dtn in ("bf16", "f16", "f32"):
        for (M, K) in ((5, 256), (64, 4096), (3, 1000)):
            x = cases.randn((M, K), 900, DT[dtn])
            qref, sref = ops_ref.per_token_quant_fp8(x)
            q, s = ops.per_token_quant_fp8(x.to(DEV))
            assert torch.equal(s.cpu(), sref), "scale = absmax/448 must be bit-exact"
            assert torch.equal(q.cpu().float(), qref.float()), "fp8 codes must be bit-exact"
    x = cases.randn((5, 256), 900, torch.bfloat16)
    w = cases.randn((48, 256), 901, torch.float32, 0.05)
    ws = w.abs().amax(1, keepdim=True) / 448.0
    wq = (w / ws).clamp(-448, 448).to(torch.float8_e4m3fn)
    xq, xs = ops.per_token_quant_fp8(x.to(DEV))
    y = ops.fp8_scaled_mm(xq, wq.to(DEV), xs, ws.to(DEV), torch.bfloat16)
    assert maxdiff(y, ops_ref.fp8_linear(x, wq, ws)) <= 8e-3
    if "fp8/fallback_scaled_mm" in golden_ops:
        assert maxdiff(y, t(golden_ops["fp8/fallback_scaled_mm"])) <= 1e-2
    for (M, N, K) in ((64, 1024, 4096), (3, 1000, 1024), (16, 512, 14336), (100, 256, 512), (1, 4096, 4096)):
        x = cases.randn((M, K), 11, torch.bfloat16)
        w = cases.randn((N, K), 12, torch.float32, 1 / math.sqrt(K))
        ws = w.abs().amax(1, keepdim=True) / 448.0
        wq = (w / ws).clamp(-448, 448).to(torch.float8_e4m3fn)
        xq, xs = ops.per_token_quant_fp8(x.to(DEV))
        y = ops.fp8_scaled_mm(xq, wq.to(DEV), xs, ws.to(DEV), torch.bfloat16)
        exact = (xq.cpu().double() @ wq.double().t()) * xs.cpu().double() * ws.double().t()
        assert maxdiff(y, exact) <= 1.6e-2, (M, N, K)


def test_fp8_packed_weight_streaming_gemm(ops):
    """fp8 W8A8 through the single-launch weight-streaming kernel (packed weight) against the
    oracle's apply_fp8_linear restatement and the row-major fp8 kernel, for split and unsplit
    plans, every M tile count, bias and fp32 output; repeats give identical bits."""
    for i, (M, N, K) in enumerate([(64, 6144, 4096), (64, 4096, 14336), (17, 28672, 4096), (1, 512, 512), (33, 1000, 1024)]):
        x = cases.randn((M, K), 400 + i, torch.bfloat16)
        w = cases.randn((N, K), 420 + i, torch.bfloat16, 1.0 / math.sqrt(K))
        b = cases.randn((N,), 440 + i, torch.bfloat16) if i % 2 else None
        from sglang_custom_amd.layers.quantization.w8a8_fp8 import quantize_weight_per_channel
        wq, ws = quantize_weight_per_channel(w)
        ref = ops_ref.fp8_linear(x, wq, ws, b)
        xq, xs = ops.per_token_quant_fp8(x.to(DEV))
        wp = ops.pack_weight_fp8(wq.to(DEV))
        bd = b.to(DEV) if b is not None else None
        got = ops.fp8_scaled_mm_packed(xq, wp, N, xs, ws.to(DEV), torch.bfloat16, bd)
        base = ops.fp8_scaled_mm(xq, wq.to(DEV), xs, ws.to(DEV), torch.bfloat16, bd)
        assert got.shape == (M, N)
        assert maxdiff(got, ref) <= ROUND_TOL["bf16"], (M, N, K)
        assert maxdiff(got, base) <= ROUND_TOL["bf16"], (M, N, K)
        assert (got != base).float().mean() < 0.01       # fp32 sums in another order: rare flips
        assert torch.equal(got, ops.fp8_scaled_mm_packed(xq, wp, N, xs, ws.to(DEV), torch.bfloat16, bd))
        g32 = ops.fp8_scaled_mm_packed(xq, wp, N, xs, ws.to(DEV), torch.float32, bd.float() if bd is not None else None)
        assert maxdiff(g32, got) <= ROUND_TOL["bf16"]


def test_sampling(ops, golden_ops):
    for i in range(len(cases.SAMPLING_CASES)):
        logits, temps, ks, ps, mps = cases.sampling_inputs(i)
        assert torch.equal(ops.argmax(logits.to(DEV)).cpu(), t(golden_ops[f"sampling/{i}/argmax"]))
        probs = ops.softmax_temperature_(logits.clone().to(DEV), temps.view(-1).to(DEV))
        pref = ops_ref.softmax_temperature(logits, temps)
        assert maxdiff(probs, pref) <= 1e-6
        # draws land in the reference's kept set, for both filter variants
        for use_minp in (False, True):
            dist = ops_ref.kept_distribution(pref, ks, ps, mps if use_minp else None)
            for trial in range(8):
                u = torch.rand(logits.shape[0], generator=torch.Generator().manual_seed(trial))
                ids = ops.top_k_top_p_min_p_sample(pref.to(DEV), ks.to(DEV), ps.to(DEV),
                                                   mps.to(DEV) if use_minp else None, u.to(DEV)).cpu()
                assert (dist.gather(1, ids.long()[:, None]) > 0).all()
    # ties: first maximal index wins (torch.argmax rule)
    lg = torch.zeros(3, 1000)
    lg[0, 7] = lg[0, 500] = 5.0
    lg[1, 999] = 1.0
    assert ops.argmax(lg.to(DEV)).tolist() == [7, 999, 0]


def test_sampling_distribution_matches_reference_sampler(ops):
    """Statistical check in the style of sgl-kernel/tests/test_sampling.py: empirical frequencies
    of the HIP sampler follow the reference sampler's renormalised distribution."""
    V, n = 500, 20000
    g = torch.Generator().manual_seed(3)
    probs = torch.softmax(2 * torch.randn((1, V), generator=g), -1)
    ks, ps = torch.tensor([25], dtype=torch.int32), torch.tensor([0.85])
    dist = ops_ref.kept_distribution(probs, ks, ps)[0]
    u = torch.rand(n, generator=g)
    ids = ops.top_k_top_p_min_p_sample(probs.expand(n, V).contiguous().to(DEV), ks.expand(n).contiguous().to(DEV),
                                       ps.expand(n).contiguous().to(DEV), None, u.to(DEV)).cpu()
    emp = torch.bincount(ids.long(), minlength=V).float() / n
    assert (emp[dist == 0] == 0).all()
    assert (emp - dist).abs().sum() < 0.05
    # same uniforms => same draws (deterministic given u)
    ids2 = ops.top_k_top_p_min_p_sample(probs.expand(n, V).contiguous().to(DEV), ks.expand(n).contiguous().to(DEV),
                                        ps.expand(n).contiguous().to(DEV), None, u.to(DEV)).cpu()
    assert torch.equal(ids, ids2)


def test_errors_are_reported_not_thrown_by_the_library(ops):
    from sglang_custom_amd._lib import Mi355Error

    x = torch.randn(4, 64, device=DEV).to(torch.float64)
    with pytest.raises(TypeError):
        ops.rmsnorm(x, torch.ones(64, device=DEV, dtype=torch.float64), 1e-5)
    q = torch.randn(2, 8, 64, device=DEV)
    kb = torch.randn(10, 3, 64, device=DEV)
    with pytest.raises(Mi355Error):  # 8 q heads are not a multiple of 3 kv heads
        ops.decode_attention_paged(q, kb, kb, torch.empty_like(q), torch.zeros(2, 8, dtype=torch.int32, device=DEV),
                                   torch.zeros(2, dtype=torch.int64, device=DEV), torch.ones(2, dtype=torch.int64, device=DEV),
                                   None, None, 1, 0.125)
    with pytest.raises(RuntimeError):
        ops.silu_and_mul(torch.randn(2, 8))  # CPU tensor: no fallback


@pytest.mark.gpu
def test_prefetch_hint_rides_rmsnorm_without_changing_results(ops):
    """mi355_prefetch_next: the next row-norm launch carries L2 warm-up workgroups for a packed
    weight.  Outputs of the norm and of the GEMM behind it are bit-identical with and without the
    hint, for served shapes, ignored shapes (split-K plans, K not a multiple of 256) and ragged N."""
    torch.manual_seed(0)
    for (m, n, k) in [(64, 28672, 4096), (7, 28672, 4096), (64, 4096, 4096), (16, 1200, 512), (3, 9000, 2048)]:
        w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.05
        wp = ops.pack_weight(w)
        x = torch.randn(m, 4096, device="cuda", dtype=torch.bfloat16)
        res = torch.randn_like(x)
        g = torch.randn(4096, device="cuda", dtype=torch.bfloat16)
        xin = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
        x0, r0 = x.clone(), res.clone()
        ops.fused_add_rmsnorm(x0, r0, g, 1e-5)
        y0 = ops.linear_packed(xin, wp, n)
        n0 = ops.rmsnorm(x, g, 1e-5)
        x1, r1 = x.clone(), res.clone()
        ops.prefetch_next(wp, m, n, k)
        ops.fused_add_rmsnorm(x1, r1, g, 1e-5)
        y1 = ops.linear_packed(xin, wp, n)
        ops.prefetch_next(wp, m, n, k, budget_bytes=1 << 30)      # more than the weight holds: clamped
        n1 = ops.rmsnorm(x, g, 1e-5)
        torch.cuda.synchronize()
        assert torch.equal(x0, x1) and torch.equal(r0, r1) and torch.equal(y0, y1) and torch.equal(n0, n1)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_deferred_splitk_reduction_is_bit_identical(ops, dtype):
    """linear_packed_partials + fused_add_rmsnorm_partials == linear_packed + fused_add_rmsnorm, bit for
    bit (same split order, one rounding), over split and unsplit plans, ragged M and N % 64 != 0."""
    torch.manual_seed(1)
    for (m, n, k) in [(64, 4096, 4096), (64, 4096, 14336), (5, 4096, 4096), (33, 2048, 8192), (16, 1208, 512), (1, 4096, 256)]:
        w = torch.randn(n, k, device="cuda", dtype=dtype) * 0.05
        wp = ops.pack_weight(w)
        x = torch.randn(m, k, device="cuda", dtype=dtype)
        res = torch.randn(m, n, device="cuda", dtype=dtype)
        g = torch.randn(n, device="cuda", dtype=dtype)
        y = ops.linear_packed(x, wp, n)
        r0 = res.clone()
        ops.fused_add_rmsnorm(y, r0, g, 1e-5)
        part = ops.linear_packed_partials(x, wp, n)
        assert part is not None and part.shape[1:] == (m, n)
        r1 = res.clone()
        y1 = ops.fused_add_rmsnorm_partials(part, r1, g, 1e-5)
        torch.cuda.synchronize()
        assert torch.equal(r0, r1), (m, n, k, part.shape[0])
        assert torch.equal(y, y1), (m, n, k, part.shape[0])
    assert ops.linear_packed_partials(torch.randn(4, 96, device="cuda", dtype=dtype),
                                      ops.pack_weight(torch.randn(64, 96, device="cuda", dtype=dtype)), 64) is None


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,cache_dtype", [(torch.bfloat16, torch.float32), (torch.float16, torch.float16)])
def test_fused_qkv_decode_attention_is_bit_identical(ops, dtype, cache_dtype):
    """qkv partials -> (reduce + RoPE + KV write + attention in one launch) against the unfused
    sequence linear_packed -> rope_inplace(with KV write) -> decode_attention_paged: outputs and both
    caches bit-identical; ragged lengths (new key in every row / wave position of a tile), GQA 1/2/4/8,
    split-KV on and off; padded rows (seq_len 1, sink slot) leave the real slots untouched."""
    torch.manual_seed(2)
    D, hidden = 128, 1024
    for hq, hkv, bs, maxlen, splits in [(32, 8, 9, 300, 1), (8, 8, 5, 70, 1), (16, 2, 6, 530, 3), (8, 4, 4, 129, 2), (8, 1, 19, 90, 1)]:
        n = (hq + 2 * hkv) * D
        w = torch.randn(n, hidden, device="cuda", dtype=dtype) * 0.04
        wp = ops.pack_weight(w)
        x = torch.randn(bs, hidden, device="cuda", dtype=dtype)
        lens = torch.randint(1, maxlen, (bs,), device="cuda", dtype=torch.int64)
        lens[0] = maxlen                     # a full-length row
        lens[-1] = 1                         # and a padded-looking one
        for i in range(1, min(bs - 1, 6)):   # newest key on different tile rows / waves
            lens[i] = min(17 * i + 1, maxlen)   # newest key = row i of tile i (wave i & 3)
        slots = int(lens.sum().item()) + 8
        perm = (torch.randperm(slots - 1, device="cuda") + 1).to(torch.int32)
        r2t = torch.zeros(bs + 3, maxlen + 8, dtype=torch.int32, device="cuda")
        rpi = torch.randperm(bs + 3, device="cuda")[:bs].to(torch.int64)
        off = 0
        for b in range(bs):
            L = int(lens[b])
            r2t[rpi[b], :L] = perm[off:off + L]
            off += L
        loc = torch.stack([r2t[rpi[b], int(lens[b]) - 1] for b in range(bs)]).to(torch.int64)
        loc[-1] = 0                          # padded row: writes go to the sink slot, reads are replaced in registers
        positions = lens - 1
        cache = torch.cat([torch.randn(maxlen + 8, D // 2, device="cuda").cos(), torch.randn(maxlen + 8, D // 2, device="cuda").sin()], dim=1).to(cache_dtype).contiguous()
        kb0 = torch.randn(slots, hkv, D, device="cuda", dtype=dtype)
        vb0 = torch.randn(slots, hkv, D, device="cuda", dtype=dtype)
        logits = torch.empty(bs, hq, 8, D, dtype=torch.float32, device="cuda")
        lse = torch.empty(bs, hq, 8, dtype=torch.float32, device="cuda")
        # unfused
        kb, vb = kb0.clone(), vb0.clone()
        qkv = ops.linear_packed(x, wp, n)
        q, k, v = qkv.split([hq * D, hkv * D, hkv * D], dim=-1)
        ops.rope_inplace(positions, q, k, D, cache, True, v=v, k_buffer=kb, v_buffer=vb, loc=loc)
        o_ref = torch.empty(bs, hq, D, device="cuda", dtype=dtype)
        ops.decode_attention_paged(q.reshape(bs, hq * D).contiguous().view(bs, hq, D), kb, vb, o_ref, r2t, rpi, lens, logits, lse, splits, D ** -0.5)
        # fused
        kb2, vb2 = kb0.clone(), vb0.clone()
        part = ops.linear_packed_partials(x, wp, n)
        o = torch.empty_like(o_ref)
        ops.decode_attention_fused_qkv(part, positions, loc, cache, kb2, vb2, o, r2t, rpi, lens, logits, lse, splits, D ** -0.5)
        torch.cuda.synchronize()
        assert torch.equal(kb, kb2) and torch.equal(vb, vb2), (hq, hkv)
        # the padded row attends over its own new key only in the fused kernel; the unfused one reads the cache
        assert torch.equal(o[:-1], o_ref[:-1]), (hq, hkv, (o[:-1].float() - o_ref[:-1].float()).abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("dtn", ["bf16", "f16"])
def test_mla_decode_mfma_kernel_shapes(ops, dtn):
    """The 16-heads-per-tile MLA kernel: head counts that do not fill a tile, several tiles, ragged
    lengths around the 16-key / 64-key steps, split-KV with empty splits; against float64."""
    rank, rope = 512, 64
    D = rank + rope
    dt = DT[dtn]
    for H, lens, splits in [(16, [1, 15, 16, 17, 63, 64, 65, 300], 1), (20, [129, 700], 4), (128, [1000, 3], 2), (5, [40], 8)]:
        total = sum(lens)
        rows = cases.randn((total + 1, 1, D), 21, dt)
        kvb = rows.to(DEV).contiguous()
        r2t = torch.zeros((len(lens), max(lens)), dtype=torch.int32)
        perm = torch.randperm(total, generator=torch.Generator().manual_seed(3)) + 1
        p = 0
        for b, n in enumerate(lens):
            r2t[b, :n] = perm[p: p + n].int()
            p += n
        rpi = torch.arange(len(lens), dtype=torch.int64)
        q = cases.randn((len(lens), H, D), 22, dt)
        o = torch.empty((len(lens), H, rank), dtype=dt, device=DEV)
        lg = torch.empty(len(lens) * H * splits * rank, dtype=torch.float32, device=DEV)
        ls = torch.empty(len(lens) * H * splits, dtype=torch.float32, device=DEV)
        ops.decode_attention_mla(q.to(DEV), kvb, o, r2t.to(DEV), rpi.to(DEV),
                                 torch.tensor(lens, dtype=torch.int64).to(DEV), lg, ls, splits, rank, D ** -0.5)
        p = 0
        for b, n in enumerate(lens):
            kv = rows[perm[p: p + n], 0]
            ref = ops_ref.attention_fp64(q[b][:, None], kv[None], kv[None, :, :rank], D ** -0.5, None)[:, 0]
            assert maxdiff(o[b], ref) <= ATTN_TOL["bf16" if dtn == "bf16" else "f16"], (dtn, H, b, splits)
            p += n


@pytest.mark.gpu
def test_fp8_fused_producers_and_deferred_reduction_are_bit_identical(ops):
    """The fp8 decode fusions against the unfused launches they replace, bit for bit:
    rmsnorm_ex(quant) == (fused_add_)rmsnorm -> per_token_quant_fp8; silu_and_mul_quant_fp8 ==
    silu_and_mul -> per_token_quant_fp8; fp8_packed_partials + rmsnorm_ex(partials, scales) ==
    fp8_scaled_mm_packed -> fused_add_rmsnorm."""
    torch.manual_seed(4)
    dt = torch.bfloat16
    for m, hidden in [(64, 4096), (3, 1024), (17, 2048)]:
        x = torch.randn(m, hidden, device="cuda", dtype=dt)
        res = torch.randn(m, hidden, device="cuda", dtype=dt)
        g = torch.randn(hidden, device="cuda", dtype=dt)
        # plain norm + quant
        y0 = ops.rmsnorm(x, g, 1e-5)
        q0, s0 = ops.per_token_quant_fp8(y0)
        y1, q1, s1 = ops.rmsnorm_ex(g, 1e-5, x=x, quant_fp8=True)
        assert torch.equal(y0, y1) and torch.equal(q0.view(torch.uint8), q1.view(torch.uint8)) and torch.equal(s0, s1)
        # fused add + norm + quant
        xa, ra = x.clone(), res.clone()
        ops.fused_add_rmsnorm(xa, ra, g, 1e-5)
        qa, sa = ops.per_token_quant_fp8(xa)
        rb = res.clone()
        yb, qb, sb = ops.rmsnorm_ex(g, 1e-5, x=x, residual=rb, quant_fp8=True)
        assert torch.equal(xa, yb) and torch.equal(ra, rb)
        assert torch.equal(qa.view(torch.uint8), qb.view(torch.uint8)) and torch.equal(sa, sb)
    for m, d in [(64, 14336), (5, 1536), (1, 8)]:
        x = torch.randn(m, 2 * d, device="cuda", dtype=dt)
        q0, s0 = ops.per_token_quant_fp8(ops.silu_and_mul(x))
        q1, s1 = ops.silu_and_mul_quant_fp8(x)
        assert torch.equal(q0.view(torch.uint8), q1.view(torch.uint8)) and torch.equal(s0, s1)
    from sglang_custom_amd.layers.quantization.w8a8_fp8 import quantize_weight_per_channel
    for m, n, k in [(64, 4096, 4096), (64, 4096, 14336), (9, 2048, 1024), (33, 1208, 512)]:
        wq, wsc = quantize_weight_per_channel(torch.randn(n, k, device="cuda") * 0.05)
        wp = ops.pack_weight_fp8(wq.contiguous())
        xq, xs = ops.per_token_quant_fp8(torch.randn(m, k, device="cuda", dtype=dt))
        res = torch.randn(m, n, device="cuda", dtype=dt)
        g = torch.randn(n, device="cuda", dtype=dt)
        y = ops.fp8_scaled_mm_packed(xq, wp, n, xs, wsc.contiguous(), dt)
        r0 = res.clone()
        ops.fused_add_rmsnorm(y, r0, g, 1e-5)
        q0, s0 = ops.per_token_quant_fp8(y)
        part = ops.fp8_packed_partials(xq, wp, n)
        assert part is not None
        r1 = res.clone()
        y1, q1, s1 = ops.rmsnorm_ex(g, 1e-5, residual=r1, part=part, part_x_scale=xs, part_w_scale=wsc.contiguous(),
                                    quant_fp8=True)
        torch.cuda.synchronize()
        assert torch.equal(r0, r1) and torch.equal(y, y1), (m, n, k, part.shape[0])
        assert torch.equal(q0.view(torch.uint8), q1.view(torch.uint8)) and torch.equal(s0, s1)


@pytest.mark.gpu
@pytest.mark.parametrize("dtn", ["bf16", "f16"])
def test_fp8_kv_cache_write_decode_extend_and_fused(ops, dtn):
    """kv_cache_dtype fp8_e4m3: set_kv_buffer stores clamp-round-to-e4m3 bytes (== torch's cast), decode
    and prefix-extend attention over the fp8 cache equal the same kernels over a 16-bit cache holding
    the widened values bit for bit, and the fused decode launch equals the unfused fp8 sequence."""
    dt = DT[dtn]
    torch.manual_seed(8)
    Hq, Hkv, D = 8, 2, 128
    lens = [70, 1, 33, 129, 16]
    B, total = len(lens), sum(lens)
    slots = total + 5
    k = torch.randn(total, Hkv, D, device=DEV, dtype=dt) * 3
    v = torch.randn(total, Hkv, D, device=DEV, dtype=dt) * 3
    k[0, 0, :4] = torch.tensor([500.0, -1000.0, 448.0, 1e-4], device=DEV, dtype=dt)     # saturation / underflow
    loc = (torch.randperm(slots - 1, device=DEV)[:total] + 1)
    kb8 = torch.zeros(slots, Hkv, D, device=DEV, dtype=torch.float8_e4m3fn)
    vb8 = torch.zeros_like(kb8)
    ops.set_kv_buffer(kb8, vb8, loc, k, v)
    ref_k = ops_ref.kv_to_fp8_and_back(k.cpu())
    assert torch.equal(kb8[loc].view(torch.uint8).cpu(), k.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).cpu())
    kbw, vbw = kb8.to(dt), vb8.to(dt)                      # the widened cache
    assert torch.equal(kbw[loc].cpu(), ref_k)
    r2t = torch.zeros(B, max(lens), dtype=torch.int32, device=DEV)
    off = 0
    for b, n in enumerate(lens):
        r2t[b, :n] = loc[off:off + n].int()
        off += n
    rpi = torch.arange(B, device=DEV)
    sl = torch.tensor(lens, device=DEV, dtype=torch.int64)
    q = torch.randn(B, Hq, D, device=DEV, dtype=dt)
    for splits in (1, 3):
        lg = torch.empty(B, Hq, splits, D, dtype=torch.float32, device=DEV)
        ls = torch.empty(B, Hq, splits, dtype=torch.float32, device=DEV)
        o8, ow = torch.empty_like(q), torch.empty_like(q)
        ops.decode_attention_paged(q, kb8, vb8, o8, r2t, rpi, sl, lg, ls, splits, D ** -0.5)
        ops.decode_attention_paged(q, kbw, vbw, ow, r2t, rpi, sl, lg, ls, splits, D ** -0.5)
        assert torch.equal(o8, ow), splits
    # extend over an fp8 prefix: the last `ext` tokens of every sequence attend over prefix + themselves
    ext = [min(n, 9) for n in lens]
    pre = [n - e for n, e in zip(lens, ext)]
    rows = [torch.arange(sum(lens[:b]) + pre[b], sum(lens[:b]) + lens[b], device=DEV) for b in range(B)]
    qe = torch.randn(sum(ext), Hq, D, device=DEV, dtype=dt)
    ke, ve = kbw[loc][torch.cat(rows)].contiguous(), vbw[loc][torch.cat(rows)].contiguous()
    qo = torch.tensor([0] + list(np.cumsum(ext)), dtype=torch.int32, device=DEV)
    kvp = torch.tensor([0] + list(np.cumsum(pre)), dtype=torch.int32, device=DEV)
    kvi = torch.cat([r2t[b, :pre[b]] for b in range(B)] + [torch.zeros(1, dtype=torch.int32, device=DEV)])
    oe8, oew = torch.empty_like(qe), torch.empty_like(qe)
    ops.extend_attention_fwd(qe, ke, ve, oe8, kb8, vb8, qo, kvp, kvi, None, True, None, max(ext), D ** -0.5)
    ops.extend_attention_fwd(qe, ke, ve, oew, kbw, vbw, qo, kvp, kvi, None, True, None, max(ext), D ** -0.5)
    assert torch.equal(oe8, oew)
    # fused decode launch over the fp8 cache == linear -> rope -> set_kv (fp8) -> decode attention
    hidden = 512
    n = (Hq + 2 * Hkv) * D
    w = torch.randn(n, hidden, device=DEV, dtype=dt) * 0.04
    wp = ops.pack_weight(w)
    x = torch.randn(B, hidden, device=DEV, dtype=dt)
    newloc = torch.stack([r2t[b, lens[b] - 1] for b in range(B)]).to(torch.int64)
    cache = torch.cat([torch.randn(200, D // 2, device=DEV).cos(), torch.randn(200, D // 2, device=DEV).sin()], 1).contiguous()
    lg = torch.empty(B, Hq, 1, D, dtype=torch.float32, device=DEV)
    ls = torch.empty(B, Hq, 1, dtype=torch.float32, device=DEV)
    ka, va = kb8.clone(), vb8.clone()
    qkv = ops.linear_packed(x, wp, n)
    q2, k2, v2 = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    ops.rope_inplace(sl - 1, q2, k2, D, cache, True)
    ops.set_kv_buffer(ka, va, newloc, k2.reshape(B, Hkv, D), v2.reshape(B, Hkv, D))
    o_ref = torch.empty(B, Hq, D, device=DEV, dtype=dt)
    ops.decode_attention_paged(q2.reshape(B, Hq * D).contiguous().view(B, Hq, D), ka, va, o_ref, r2t, rpi, sl, lg, ls, 1, D ** -0.5)
    kf, vf = kb8.clone(), vb8.clone()
    o = torch.empty_like(o_ref)
    ops.decode_attention_fused_qkv(ops.linear_packed_partials(x, wp, n), sl - 1, newloc, cache, kf, vf, o, r2t, rpi, sl,
                                   lg, ls, 1, D ** -0.5)
    torch.cuda.synchronize()
    assert torch.equal(ka.view(torch.uint8), kf.view(torch.uint8)) and torch.equal(va.view(torch.uint8), vf.view(torch.uint8))
    assert torch.equal(o, o_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("dtn", ["bf16", "f16"])
@pytest.mark.parametrize("Hq,Hkv,D,window,cap", [(8, 2, 64, -1, 0.0), (4, 2, 256, -1, 0.0), (6, 3, 80, -1, 0.0),
                                                 (8, 2, 128, 24, 0.0), (8, 2, 128, -1, 30.0)])
def test_fp8_kv_cache_outside_the_matrix_core_shapes(ops, dtn, Hq, Hkv, D, window, cap):
    """An fp8 cache at head dims other than 128 (Llama-3.2-1B has 64), or under a sliding window / logit cap:
    the write is torch's cast, decode agrees with the oracle on the widened values (bit for bit with the 16-bit
    cache where one kernel serves both), prefix-extend attention equals the same call over a 16-bit cache holding
    the widened values bit for bit (the reference's pool stores fp8 for any model, memory_pool.py:436-446)."""
    dt = DT[dtn]
    torch.manual_seed(D + Hq)
    lens = [70, 1, 33, 129, 16]
    B, total = len(lens), sum(lens)
    slots = total + 5
    k = torch.randn(total, Hkv, D, device=DEV, dtype=dt) * 3
    v = torch.randn(total, Hkv, D, device=DEV, dtype=dt) * 3
    loc = (torch.randperm(slots - 1, device=DEV)[:total] + 1)
    kb8 = torch.zeros(slots, Hkv, D, device=DEV, dtype=torch.float8_e4m3fn)
    vb8 = torch.zeros_like(kb8)
    ops.set_kv_buffer(kb8, vb8, loc, k, v)
    assert torch.equal(kb8[loc].view(torch.uint8), k.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8))
    kbw, vbw = kb8.to(dt), vb8.to(dt)
    r2t = torch.zeros(B, max(lens), dtype=torch.int32, device=DEV)
    off = 0
    for b, n in enumerate(lens):
        r2t[b, :n] = loc[off:off + n].int()
        off += n
    rpi = torch.arange(B, device=DEV)
    sl = torch.tensor(lens, device=DEV, dtype=torch.int64)
    q = torch.randn(B, Hq, D, device=DEV, dtype=dt)
    for splits in (1, 3):
        lg = torch.empty(B, Hq, splits, D, dtype=torch.float32, device=DEV)
        ls = torch.empty(B, Hq, splits, dtype=torch.float32, device=DEV)
        o8, ow = torch.empty_like(q), torch.empty_like(q)
        ops.decode_attention_paged(q, kb8, vb8, o8, r2t, rpi, sl, lg, ls, splits, D ** -0.5, logit_cap=cap)
        ops.decode_attention_paged(q, kbw, vbw, ow, r2t, rpi, sl, lg, ls, splits, D ** -0.5, logit_cap=cap)
        if D == 128:    # the same kernel serves both caches: same bits
            assert torch.equal(o8, ow), splits
        else:           # the fp8 cache goes through the wave-per-head kernel, the 16-bit one through the vector kernel
            assert (o8.float() - ow.float()).abs().max().item() <= 2e-2 * max(1.0, ow.float().abs().max().item()), splits
        ref = ops_ref.decode_attention(q.cpu().float(), kbw.cpu().float(), vbw.cpu().float(), r2t.cpu(), rpi.cpu(), sl.cpu(),
                                       D ** -0.5, Hq != Hkv) if cap == 0.0 else None
        if ref is not None:
            assert (o8.cpu().float() - ref).abs().max().item() < 6e-2
    ext = [min(n, 9) for n in lens]
    pre = [n - e for n, e in zip(lens, ext)]
    rows = [torch.arange(sum(lens[:b]) + pre[b], sum(lens[:b]) + lens[b], device=DEV) for b in range(B)]
    qe = torch.randn(sum(ext), Hq, D, device=DEV, dtype=dt)
    ke, ve = kbw[loc][torch.cat(rows)].contiguous(), vbw[loc][torch.cat(rows)].contiguous()
    qo = torch.tensor([0] + list(np.cumsum(ext)), dtype=torch.int32, device=DEV)
    kvp = torch.tensor([0] + list(np.cumsum(pre)), dtype=torch.int32, device=DEV)
    kvi = torch.cat([r2t[b, :pre[b]] for b in range(B)] + [torch.zeros(1, dtype=torch.int32, device=DEV)])
    oe8, oew = torch.empty_like(qe), torch.empty_like(qe)
    ops.extend_attention_fwd(qe, ke, ve, oe8, kb8, vb8, qo, kvp, kvi, None, True, None, max(ext), D ** -0.5, cap,
                             sliding_window_size=window)
    ops.extend_attention_fwd(qe, ke, ve, oew, kbw, vbw, qo, kvp, kvi, None, True, None, max(ext), D ** -0.5, cap,
                             sliding_window_size=window)
    torch.cuda.synchronize()
    assert torch.equal(oe8, oew)
    assert oe8.float().abs().max().item() > 0


def _extend_kernel_case(ops, dt, Hq, Hkv, D, pre, ext, window, cap, seed, tol):
    """mi355_extend_attention with the prefix lists a sliding-window layer hands it, against the
    float64 statement of the reference kernel's masking rules."""
    seq = [a + b for a, b in zip(pre, ext)]
    B = len(seq)
    kc, vc, r2t, pool = cases.paged_inputs(seed, dt, B, Hkv, D, seq)
    q = cases.randn((sum(ext), Hq, D), seed + 1, dt)
    wins = [ops_ref.window_of(p_, window) if window > 0 else (0, p_) for p_ in pre]
    kvp = torch.zeros(B + 1, dtype=torch.int32)
    kvp[1:] = torch.cumsum(torch.tensor([n for _, n in wins]), 0)
    kvi = torch.cat([r2t[pool[b], wins[b][0]: wins[b][0] + wins[b][1]] for b in range(B)] + [torch.zeros(1, dtype=torch.int32)])
    kext = torch.cat([kc[r2t[pool[b], pre[b]: seq[b]].long()] for b in range(B)])
    vext = torch.cat([vc[r2t[pool[b], pre[b]: seq[b]].long()] for b in range(B)])
    qo = torch.zeros(B + 1, dtype=torch.int32)
    qo[1:] = torch.cumsum(torch.tensor(ext), 0)
    o = torch.empty((sum(ext), Hq, D), dtype=dt, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), kext.to(DEV), vext.to(DEV), o, kc.to(DEV), vc.to(DEV), qo.to(DEV), kvp.to(DEV),
                             kvi.to(DEV), None, True, None, max(ext), D ** -0.5, logit_cap=cap,
                             sliding_window_size=window)
    p = 0
    for b in range(B):
        toks = r2t[pool[b], wins[b][0]: wins[b][0] + wins[b][1]].long()
        ref = ops_ref.extend_attention_kernel_rules_fp64(q[p: p + ext[b]], kc[toks], vc[toks], kext[p: p + ext[b]],
                                                         vext[p: p + ext[b]], D ** -0.5, True, cap, window)
        assert maxdiff(o[p: p + ext[b]], ref) <= tol, (Hq, Hkv, D, window, cap, b)
        p += ext[b]


@pytest.mark.parametrize("dtn", ["f32", "bf16"])
def test_extend_attention_sliding_window_and_logit_cap(ops, dtn):
    """Sliding-window prefix lists + the in-kernel window mask, and the tanh logit cap, alone and
    together; windows shorter and longer than the prefix and than the new-token count (the window
    term can cut keys only when the new tokens outnumber window: row i loses prefix key k < i - window)."""
    dt = DT[dtn]
    for (Hq, Hkv, D), (pre, ext) in [((8, 2, 128), ([40, 0, 13, 300], [30, 70, 3, 129])), ((12, 4, 64), ([7, 90], [100, 1]))]:
        for window, cap in [(16, 0.0), (64, 0.0), (-1, 30.0), (16, 50.0), (1000, 0.0)]:
            _extend_kernel_case(ops, dt, Hq, Hkv, D, pre, ext, window, cap, 140, ATTN_TOL[dtn])


def test_backend_sliding_window_layers_follow_the_reference_window_lists(ops):
    """Mi355AttnBackend on a model with sliding_window_size: an extend over cached prefixes reads the
    last min(prefix, window + 1) prefix tokens (plus the in-kernel window mask), a decode the last
    min(len, window + 1) tokens -- layers without a window in the same model read everything."""
    from types import SimpleNamespace

    from sglang_custom_amd.layers.attention.mi355_backend import Mi355AttnBackend
    from sglang_custom_amd.layers.radix_attention import RadixAttention
    from sglang_custom_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool
    from sglang_custom_amd.model_executor.forward_batch_info import ForwardBatch, ForwardMode

    dt, Hq, Hkv, D, W = torch.bfloat16, 8, 2, 128, 24
    pre, ext = [50, 0, 9], [40, 33, 1]
    seq = [a + b for a, b in zip(pre, ext)]
    kc, vc, r2t, pool_rows = cases.paged_inputs(150, dt, 3, Hkv, D, [n + 1 for n in seq])   # + one decode token each
    q = cases.randn((sum(ext), Hq, D), 151, dt)
    kvpool = MHATokenToKVPool(kc.shape[0] - 1, 1, dt, Hkv, D, 1, DEV)
    kvpool.get_key_buffer(0).copy_(kc.to(DEV))
    kvpool.get_value_buffer(0).copy_(vc.to(DEV))
    r2tp = ReqToTokenPool(r2t.shape[0], r2t.shape[1], DEV)
    r2tp.req_to_token.copy_(r2t.to(DEV))
    runner = SimpleNamespace(device=DEV, req_to_token_pool=r2tp, token_to_kv_pool=kvpool, tp_size=1, sliding_window_size=W,
                             model_config=SimpleNamespace(num_attention_heads=Hq, num_key_value_heads=Hkv))
    be = Mi355AttnBackend(runner)
    win_layer = RadixAttention(Hq, D, D ** -0.5, Hkv, 0, sliding_window_size=W)
    full_layer = RadixAttention(Hq, D, D ** -0.5, Hkv, 0)
    kext = torch.cat([kc[r2t[pool_rows[b], pre[b]: seq[b]].long()] for b in range(3)])
    vext = torch.cat([vc[r2t[pool_rows[b], pre[b]: seq[b]].long()] for b in range(3)])
    loc = torch.cat([r2t[pool_rows[b], pre[b]: seq[b]].long() for b in range(3)])
    fb = ForwardBatch(forward_mode=ForwardMode.EXTEND, batch_size=3, input_ids=torch.zeros(sum(ext), dtype=torch.int64, device=DEV),
                      req_pool_indices=pool_rows.to(DEV), seq_lens=torch.tensor(seq, device=DEV), out_cache_loc=loc.to(DEV),
                      seq_lens_sum=sum(seq), seq_lens_cpu=torch.tensor(seq), extend_seq_lens=torch.tensor(ext, dtype=torch.int32, device=DEV),
                      extend_prefix_lens=torch.tensor(pre, dtype=torch.int32, device=DEV), extend_prefix_lens_cpu=pre,
                      extend_seq_lens_cpu=ext, req_to_token_pool=r2tp, token_to_kv_pool=kvpool, attn_backend=be)
    be.init_forward_metadata(fb)
    for layer, window in ((win_layer, W), (full_layer, -1)):
        o = be.forward_extend(q.to(DEV).view(sum(ext), -1), kext.to(DEV), vext.to(DEV), layer, fb).view(-1, Hq, D)
        p = 0
        for b in range(3):
            s0, n = ops_ref.window_of(pre[b], W) if window > 0 else (0, pre[b])
            toks = r2t[pool_rows[b], s0: s0 + n].long()
            ref = ops_ref.extend_attention_kernel_rules_fp64(q[p: p + ext[b]], kc[toks], vc[toks], kext[p: p + ext[b]],
                                                             vext[p: p + ext[b]], D ** -0.5, True, 0.0, window)
            assert maxdiff(o[p: p + ext[b]], ref) <= ATTN_TOL["bf16"], (window, b)
            p += ext[b]
    # decode: one more token per request (its K/V row is already in the cache)
    dseq = [n + 1 for n in seq]
    qd = cases.randn((3, Hq, D), 152, dt)
    dloc = torch.stack([r2t[pool_rows[b], seq[b]] for b in range(3)]).long()
    fb = ForwardBatch(forward_mode=ForwardMode.DECODE, batch_size=3, input_ids=torch.zeros(3, dtype=torch.int64, device=DEV),
                      req_pool_indices=pool_rows.to(DEV), seq_lens=torch.tensor(dseq, device=DEV), out_cache_loc=dloc.to(DEV),
                      seq_lens_sum=sum(dseq), seq_lens_cpu=torch.tensor(dseq), req_to_token_pool=r2tp, token_to_kv_pool=kvpool,
                      attn_backend=be)
    be.init_forward_metadata(fb)
    for layer, window in ((win_layer, W), (full_layer, -1)):
        o = be.forward_decode(qd.to(DEV).view(3, -1), kc[dloc].to(DEV), vc[dloc].to(DEV), layer, fb).view(3, Hq, D)
        for b in range(3):
            s0, n = ops_ref.window_of(dseq[b], W) if window > 0 else (0, dseq[b])
            toks = r2t[pool_rows[b], s0: s0 + n].long()
            ref = ops_ref.attention_fp64(qd[b][:, None], kc[toks].transpose(0, 1), vc[toks].transpose(0, 1), D ** -0.5)[:, 0]
            assert maxdiff(o[b], ref) <= ATTN_TOL["bf16"], (window, b)


def test_argmax_rows_of_minus_infinity_and_nan(ops):
    """Rows a grammar mask or an overflow can produce: everything -inf but a few entries, everything
    -inf (torch: index 0), -inf everywhere but NaNs in front (NaN never wins here; the first -inf
    does), a maximum that repeats in the columns one lane owns, odd vocabulary with a scalar tail."""
    for V in (128256, 4100, 4099, 37):
        g = torch.Generator().manual_seed(V)
        x = torch.randn((8, V), generator=g)
        x[0] = float("-inf"); x[0, V - 3] = -5.0; x[0, V // 2] = -5.0
        x[1] = float("-inf")
        x[2] = float("-inf"); x[2, :3] = float("nan")
        x[3, 5] = x[3, 5 + 4096 if V > 5000 else V - 1] = 9.0            # 5 and 5 + 4096 fall to the same lane
        x[4, : V // 2] = float("-inf")
        x[5] = 1.0
        x[6, V - 1] = 50.0
        want = torch.argmax(torch.nan_to_num(x, nan=float("-inf")), -1)
        want[2] = 3                                                       # first -inf behind the NaNs
        got = ops.argmax(x.to(DEV)).cpu()
        assert torch.equal(got, want), (V, got.tolist(), want.tolist())


def test_argmax_over_several_workgroups_per_row(ops):
    """mi355_argmax_chunked (few rows: a row is cut into chunks that meet in a 64-bit cell through an atomic
    max): the indices of the one-workgroup-per-row kernel for 1 .. 200 rows, ties across chunk boundaries
    (first index wins), negative maxima, the two zeros as one value, a row of -inf; the workspace comes back
    zeroed, back-to-back calls and a replayed graph share it."""
    from sglang_custom_amd import _lib

    lib = _lib.load()
    V = 128256
    g = torch.Generator().manual_seed(3)
    for rows in (1, 3, 64, 200):
        x = (2.0 * torch.randn((rows, V), generator=g)).bfloat16().float()
        x[0] = -torch.rand(V, generator=g) - 1.0                      # every value negative
        x[0, 70000] = x[0, 20000] = -0.5                                # a tie across chunks
        if rows > 2:
            x[1] = float("-inf")
            x[2] = -1.0; x[2, 100000] = 0.0; x[2, 5] = -0.0           # -0.0 at 5 and +0.0 at 100000 are equal: index 5
        xd = x.to(DEV)
        want = torch.empty(rows, dtype=torch.int64, device=DEV)
        assert lib.mi355_argmax(want.data_ptr(), xd.data_ptr(), rows, V, V, torch.cuda.current_stream().cuda_stream) == 0
        ws = torch.zeros(lib.mi355_argmax_workspace_bytes(rows), dtype=torch.uint8, device=DEV)
        for rep in range(2):
            got = torch.full((rows,), -7, dtype=torch.int64, device=DEV)
            assert lib.mi355_argmax_chunked(got.data_ptr(), xd.data_ptr(), rows, V, V, ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream().cuda_stream) == 0
            torch.cuda.synchronize()
            assert torch.equal(got, want), (rows, rep)
            assert int(ws.sum()) == 0, "workspace not handed back zeroed"
        assert torch.equal(want.cpu(), torch.argmax(x, -1))
        assert want[0].item() == 20000 and (rows < 3 or (want[1].item() == 0 and want[2].item() == 5))
    # through the wrapper, inside a replayed graph, on changing logits
    x = torch.randn((64, V), generator=g).to(DEV)
    ops.argmax(x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            out = ops.argmax(x)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        x.copy_(torch.randn((64, V), generator=g))
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, torch.argmax(x, -1))


def test_argmax_and_sampler_at_llama3_vocab(ops):
    """vocab 128256 (sgl-kernel/tests/test_sampling.py runs its cases at 111 / 32000 / 128256): arg-max
    with the first-index tie rule, softmax with temperature, and the joint top-k / top-p / min-p
    sampler -- every draw inside the oracle's kept set, renormalised mass as the oracle's."""
    V, B = 128256, 64
    g = torch.Generator().manual_seed(77)
    logits = (3.0 * torch.randn((B, V), generator=g)).bfloat16().float()     # bf16-valued: many exact ties
    logits[3, 100] = logits[3, 90000] = logits[3].max() + 1.0                 # a tie for the maximum: first index wins
    logits[5, V - 1] = logits[5].max() + 2.0                                  # the maximum in the last column
    ids = ops.argmax(logits.to(DEV))
    assert torch.equal(ids.cpu(), torch.argmax(logits, -1)) and ids[3].item() == 100 and ids[5].item() == V - 1
    temps = torch.rand((B, 1), generator=g) + 0.5
    probs = ops_ref.softmax_temperature(logits.clone(), temps)
    got = logits.clone().to(DEV)
    ops.softmax_temperature_(got, temps.to(DEV))
    assert torch.allclose(got.cpu(), probs, rtol=2e-5, atol=1e-12)    # 128256-term sums in another order
    # the sampler on untied probabilities (among exactly equal probabilities WHICH of them make the
    # top-k cut is the sort's business in the reference and is not defined)
    probs = ops_ref.softmax_temperature(3.0 * torch.randn((B, V), generator=g), temps)
    top_ks = torch.randint(1, 60, (B,), generator=g).to(torch.int32)
    top_ps = torch.rand((B,), generator=g) * 0.8 + 0.1
    min_ps = torch.rand((B,), generator=g) * 0.05
    dist = ops_ref.kept_distribution(probs, top_ks, top_ps, min_ps)
    for trial in range(20):
        u = torch.rand((B,), generator=g)
        out = ops.top_k_top_p_min_p_sample(probs.to(DEV), top_ks.to(DEV), top_ps.to(DEV), min_ps.to(DEV), u.to(DEV))
        assert (dist.gather(1, out.cpu().long()[:, None]) > 0).all(), trial


@pytest.mark.gpu
def test_final_logit_softcap(ops):
    """fused_softcap (logits_processor.py:579-618) at Gemma-2's cap (30) and a small one, on logits from
    tiny to far past the cap, sizes with and without a 4-element tail: the restated arithmetic to a few
    ulp, fp64 tanh to 1e-6 of the cap, finite where e^{2x} would overflow; then through LogitsProcessor
    (config.final_logit_softcapping; negative = off, logits_processor.py:217-224)."""
    from types import SimpleNamespace

    from sglang_custom_amd.layers.logits_processor import LogitsProcessor
    from sglang_custom_amd.model_executor.forward_batch_info import ForwardMode

    g = torch.Generator().manual_seed(5)
    for cap, shape in ((30.0, (64, 128256)), (30.0, (3, 1001)), (2.5, (7, 333))):
        x = torch.randn(shape, generator=g) * 25.0
        x.view(-1)[:6] = torch.tensor([0.0, 1e-7, -1e-7, 1e4, -1e4, 3.4e38])
        got = ops.softcap_inplace(x.clone().to(DEV), cap).cpu()
        assert torch.isfinite(got).all()
        assert (got - ops_ref.softcap(x, cap)).abs().max().item() <= 4e-6 * cap
        assert (got.double() - cap * torch.tanh(x.double() / cap)).abs().max().item() <= 1e-6 * cap
    # a NaN logit stays NaN (the reference's kernel and the oracle keep it; a barrier timeout of the custom
    # all-reduce poisons its output with NaN and that poison must not turn into a valid +cap logit); infinities saturate
    x = torch.tensor([float("nan"), 1.0, float("inf"), float("-inf"), float("nan"), 2.0, -3.0])
    got = ops.softcap_inplace(x.clone().to(DEV), 30.0).cpu()
    want = ops_ref.softcap(x, 30.0)
    assert torch.isnan(want[[0, 4]]).all() and torch.isnan(got[[0, 4]]).all()
    assert got[2].item() == 30.0 and got[3].item() == -30.0
    assert (got[[1, 5, 6]] - want[[1, 5, 6]]).abs().max().item() <= 4e-6 * 30
    V, H = 1000, 64
    w = (torch.randn((V, H), generator=g) * 0.5).to(DEV)
    h = (torch.randn((5, H), generator=g) * 4.0).to(DEV)
    fb = SimpleNamespace(forward_mode=ForwardMode.DECODE)
    outs = {}
    for cap in (None, -1.0, 20.0):
        lp = LogitsProcessor(SimpleNamespace(vocab_size=V, final_logit_softcapping=cap))
        outs[cap] = lp(None, h, SimpleNamespace(weight=w), fb).next_token_logits.cpu()
    assert torch.equal(outs[None], outs[-1.0])
    assert outs[None].abs().max().item() > 20.0           # the cap bites
    assert (outs[20.0] - ops_ref.softcap(outs[None], 20.0)).abs().max().item() <= 1e-4


@pytest.mark.parametrize("m,k", [(64, 2304), (3, 2304), (1, 64), (17, 1000)])
def test_fp8_rows_padded_for_the_packed_gemm(ops, m, k):
    """A reduction length the packed fp8 GEMM does not stream (K % 512 != 0, e.g. DeepSeek-V3's down_proj shard
    of 2304) is rounded up: the quantisers write rows of k_padded bytes with zeros behind the K codes, the weight
    is packed with zero columns -- same codes, same scales, and the GEMM equals the unpadded row-major kernel."""
    from sglang_custom_amd.layers.quantization.w8a8_fp8 import quantize_weight_per_channel
    g = torch.Generator().manual_seed(m * 1000 + k)
    kp = -(-k // 512) * 512
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
    q0, s0 = ops.per_token_quant_fp8(x)
    q1, s1 = ops.per_token_quant_fp8(x, pad_to=kp)
    assert q1.shape == (m, kp) and torch.equal(s0, s1)
    assert torch.equal(q1[:, :k].view(torch.uint8), q0.view(torch.uint8)) and not q1[:, k:].view(torch.uint8).any()
    if k % 8 == 0:
        gu = torch.randn(m, 2 * k, generator=g).to(torch.bfloat16).to(DEV)
        a0, t0 = ops.silu_and_mul_quant_fp8(gu)
        a1, t1 = ops.silu_and_mul_quant_fp8(gu, pad_to=kp)
        assert a1.shape == (m, kp) and torch.equal(t0, t1)
        assert torch.equal(a1[:, :k].view(torch.uint8), a0.view(torch.uint8)) and not a1[:, k:].view(torch.uint8).any()
    if k % 16 == 0:
        n = 256
        wq, ws = quantize_weight_per_channel(torch.randn(n, k, generator=g) * k ** -0.5)
        wq, ws = wq.to(DEV), ws.to(DEV)
        wpad = torch.zeros(n, kp, dtype=torch.uint8, device=DEV)
        wpad[:, :k] = wq.view(torch.uint8)
        got = ops.fp8_scaled_mm_packed(q1, ops.pack_weight_fp8(wpad.view(torch.float8_e4m3fn)), n, s1, ws, torch.bfloat16)
        base = ops.fp8_scaled_mm(q0, wq, s0, ws, torch.bfloat16)
        assert maxdiff(got, base) <= 2e-2 * base.float().abs().max().item()
        assert maxdiff(got, ops_ref.fp8_linear(x.cpu(), wq.cpu(), ws.cpu())) <= 1e-2 * max(1.0, base.float().abs().max().item())


def test_fp8_linear_method_pads_odd_reduction_lengths(ops):
    """W8A8Fp8LinearMethod on a [N, 2304] weight: the decode path runs the packed kernel on K rounded up to 2560
    (from a tensor and from a producer's QuantizedAct) and equals the oracle's apply_fp8_linear."""
    from types import SimpleNamespace
    from sglang_custom_amd.layers.quantization.unquant import QuantizedAct, decode_linear_deferred
    from sglang_custom_amd.layers.quantization.w8a8_fp8 import W8A8Fp8LinearMethod
    g = torch.Generator().manual_seed(5)
    n, k, m = 512, 2304, 48
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16)
    qm = W8A8Fp8LinearMethod()
    layer = SimpleNamespace(weight=torch.nn.Parameter(w.to(DEV), requires_grad=False), quant_method=qm, tp_size=1)
    qm.process_weights_after_loading(layer)
    assert layer.k_padded == 2560 and layer.weight_packed is not None and layer.weight.shape == (n, k)
    ref = ops_ref.fp8_linear(x, layer.weight.data.cpu(), layer.weight_scale.data.cpu())
    y = qm.apply(layer, x.to(DEV))
    assert maxdiff(y, ref) <= 1e-2
    gu = torch.randn(m, 2 * k, generator=g).to(torch.bfloat16).to(DEV)
    aq, asc = ops.silu_and_mul_quant_fp8(gu, pad_to=layer.k_padded)
    y2 = qm.apply(layer, QuantizedAct(aq, asc, torch.bfloat16))
    ref2 = ops_ref.fp8_linear(ops_ref.silu_and_mul(gu.cpu()), layer.weight.data.cpu(), layer.weight_scale.data.cpu())
    assert maxdiff(y2, ref2) <= 1e-2
    part = decode_linear_deferred(layer, x.to(DEV))
    assert part is not None
    summed = ((part.part.sum(0) * part.x_scale) * part.w_scale.view(1, -1)).to(torch.bfloat16)
    assert maxdiff(summed, ref) <= 1e-2


@pytest.mark.parametrize("dtn", ["bf16", "f16"])
def test_extend_attention_matrix_core_tile_192_wide_keys(ops, dtn):
    """The multi-head form of MLA prefill (deepseek_v2.py:1131-1179): query / key heads of 128 + 64 features against
    128-wide values on the 32x32 matrix-core tile -- ragged lengths around the 128-row / 64-key tile edges, with
    and without a cached prefix, against float64 attention."""
    dt = DT[dtn]
    Dq, Dv = 192, 128
    for H, (pre, ext) in [(16, ([0, 0, 0], [64, 257, 129])), (5, ([0], [300])), (16, ([70, 0, 129], [65, 1, 200]))]:
        seq = [a + b for a, b in zip(pre, ext)]
        nslots = sum(seq) + 1
        g = torch.Generator().manual_seed(H * 100 + sum(seq))
        kc = torch.randn(nslots, H, Dq, generator=g).to(dt)
        vc = torch.randn(nslots, H, Dv, generator=g).to(dt)
        perm = torch.randperm(nslots - 1, generator=g) + 1
        slots, p0 = [], 0
        for n in seq:
            slots.append(perm[p0: p0 + n])
            p0 += n
        q = torch.randn(sum(ext), H, Dq, generator=g).to(dt)
        kext = torch.cat([kc[slots[b][pre[b]:]] for b in range(len(seq))])
        vext = torch.cat([vc[slots[b][pre[b]:]] for b in range(len(seq))])
        qo = torch.tensor([0] + list(torch.tensor(ext).cumsum(0)), dtype=torch.int32)
        kvp = torch.tensor([0] + list(torch.tensor(pre).cumsum(0)), dtype=torch.int32)
        kvi = torch.cat([slots[b][: pre[b]] for b in range(len(seq))] + [torch.zeros(1, dtype=torch.int64)]).to(torch.int32)
        o = torch.full((sum(ext), H, Dv), float("nan"), dtype=dt, device=DEV)
        ops.extend_attention_fwd(q.to(DEV), kext.to(DEV), vext.to(DEV), o, kc.to(DEV), vc.to(DEV), qo.to(DEV), kvp.to(DEV),
                                 kvi.to(DEV), None, True, None, max(ext), Dq ** -0.5)
        p = 0
        for b in range(len(seq)):
            ref = ops_ref.attention_fp64(q[p: p + ext[b]].transpose(0, 1), kc[slots[b]].transpose(0, 1),
                                         vc[slots[b]].transpose(0, 1), Dq ** -0.5, pre[b]).transpose(0, 1)
            assert maxdiff(o[p: p + ext[b]], ref) <= ATTN_TOL[dtn], (H, b)
            p += ext[b]
