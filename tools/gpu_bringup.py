"""First-light check of every kernel against inline torch fp32 formulas (dev tool; the real
parity tests live in tests/ and use oracle/)."""
import math, sys, time, traceback
import torch
sys.path.insert(0, '.')
from sglang_custom_amd import ops

dev = 'cuda'
torch.manual_seed(0)
results = []
def report(name, err, tol):
    ok = err <= tol
    results.append(ok)
    print(f"{'OK ' if ok else 'BAD'} {name}: max_err={err:.3e} tol={tol:.1e}", flush=True)

def run(fn):
    try:
        fn()
    except Exception:
        results.append(False)
        traceback.print_exc()
    torch.cuda.synchronize()

def t_rmsnorm():
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        for rows, h in ((7, 4096), (3, 500), (5, 111), (2, 16384), (64, 2048)):
            x = torch.randn(rows, h, device=dev).to(dt); w = (1 + 0.1 * torch.randn(h, device=dev)).to(dt)
            r = torch.randn(rows, h, device=dev).to(dt)
            ref = (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-5) * w.float()).to(dt)
            out = ops.rmsnorm(x, w, 1e-5)
            tol = 1e-5 if dt == torch.float32 else 2e-2
            report(f"rmsnorm {dt} {rows}x{h}", (out.float() - ref.float()).abs().max().item(), tol)
            s = x.float() + r.float(); rref = s.to(dt)
            oref = (s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + 1e-5) * w.float()).to(dt)
            x2, r2 = x.clone(), r.clone()
            ops.fused_add_rmsnorm(x2, r2, w, 1e-5)
            report(f"fused_add_rmsnorm {dt} {rows}x{h}", max((x2.float() - oref.float()).abs().max().item(), (r2.float() - rref.float()).abs().max().item()), tol)

def t_silu():
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        for rows, d in ((5, 14336), (3, 100), (64, 1024)):
            x = torch.randn(rows, 2 * d, device=dev).to(dt)
            ref = torch.nn.functional.silu(x[:, :d]) * x[:, d:]
            out = ops.silu_and_mul(x)
            report(f"silu_and_mul {dt} {rows}x{d}", (out.float() - ref.float()).abs().max().item(), 1e-5 if dt == torch.float32 else 2e-2)

def rope_ref(pos, q, k, hd, cache, neox):
    cs = cache.index_select(0, pos); cos, sin = cs.chunk(2, -1)
    def app(x):
        T = x.shape[0]; x = x.view(T, -1, hd); rd = cache.shape[1]
        xr, xp = x[..., :rd], x[..., rd:]
        c = cos.unsqueeze(-2).to(x.dtype); s = sin.unsqueeze(-2).to(x.dtype)
        if neox: x1, x2 = torch.chunk(xr, 2, -1)
        else: x1, x2 = xr[..., ::2], xr[..., 1::2]
        o1 = x1 * c - x2 * s; o2 = x2 * c + x1 * s
        o = torch.cat((o1, o2), -1) if neox else torch.stack((o1, o2), -1).flatten(-2)
        return torch.cat((o, xp), -1).reshape(T, -1)
    return app(q), app(k)

def t_rope():
    for dt in (torch.float32, torch.bfloat16):
        for neox in (True, False):
            for hd, rd, hq, hkv in ((128, 128, 32, 8), (64, 64, 8, 2), (128, 64, 4, 4)):
                T = 37
                inv = 1.0 / (10000 ** (torch.arange(0, rd, 2, dtype=torch.float) / rd))
                t = torch.arange(512, dtype=torch.float); fr = torch.einsum('i,j->ij', t, inv)
                cache = torch.cat((fr.cos(), fr.sin()), -1).to(dev).to(dt)
                pos = torch.randint(0, 512, (T,), device=dev)
                qkv = torch.randn(T, (hq + 2 * hkv) * hd, device=dev).to(dt)
                q, k, v = qkv.split([hq * hd, hkv * hd, hkv * hd], -1)
                qr, kr = rope_ref(pos, q.clone(), k.clone(), hd, cache, neox)
                kb = torch.zeros(100, hkv, hd, device=dev, dtype=dt); vb = torch.zeros_like(kb)
                loc = torch.randperm(99, device=dev)[:T] + 1
                ops.rope_inplace(pos, q, k, hd, cache, neox, v=v, k_buffer=kb, v_buffer=vb, loc=loc)
                e = max((q.float() - qr.float()).abs().max().item(), (k.float() - kr.float()).abs().max().item(),
                        (kb[loc].view(T, -1).float() - kr.float()).abs().max().item(), (vb[loc].view(T, -1) - v).abs().max().item())
                report(f"rope {dt} neox={neox} hd={hd} rd={rd}", e, 0.0 if dt != torch.float32 else 1e-6)

def t_index():
    r2t = torch.randint(1, 1000, (64, 300), device=dev, dtype=torch.int32)
    pool = torch.randperm(64, device=dev)[:9]
    lens = torch.randint(1, 300, (9,), device=dev)
    indptr = torch.zeros(10, dtype=torch.int32, device=dev); indptr[1:] = torch.cumsum(lens, 0)
    out = torch.empty(int(lens.sum()), dtype=torch.int32, device=dev)
    ops.create_kv_indices(r2t, pool, lens, indptr, None, out)
    ref = torch.cat([r2t[pool[i], :lens[i]] for i in range(9)])
    report("create_kv_indices", float((out != ref).sum().item()), 0)
    pre = torch.randint(0, 50, (9,), device=dev); ext = torch.randint(1, 100, (9,), device=dev); seq = pre + ext
    ocl = torch.randint(1, 100000, (int(ext.sum()),), device=dev)
    r2 = r2t.clone(); ops.write_req_to_token(r2, pool, pre, seq, ext, ocl)
    r3 = r2t.clone(); pt = 0
    for i in range(9):
        r3[pool[i], pre[i]:seq[i]] = ocl[pt:pt + ext[i]].int(); pt += int(ext[i])
    report("write_req_to_token", float((r2 != r3).sum().item()), 0)
    p, s = ops.compute_position(pre.int(), ext.int(), int(ext.sum()))
    pref = torch.cat([torch.arange(int(pre[i]), int(pre[i] + ext[i]), device=dev) for i in range(9)])
    sref = torch.zeros(9, dtype=torch.int32, device=dev); sref[1:] = torch.cumsum(ext[:-1], 0)
    report("compute_position", float((p != pref).sum().item() + (s != sref).sum().item()), 0)
    sl = seq.clone(); pos = torch.empty(9, dtype=torch.int64, device=dev); oc = torch.randint(1, 9999, (9,), device=dev)
    r4 = r2.clone(); ops.prepare_decode(r4, pool, sl, oc, pos)
    r5 = r2.clone(); r5[pool, seq] = oc.int()
    report("prepare_decode", float((r4 != r5).sum().item() + (sl != seq + 1).sum().item() + (pos != seq).sum().item()), 0)

def attn_ref(q, K, V, scale, cap=0.0):
    # q [Hq, Lq, D] K [Hkv, S, D] V [Hkv, S, Dv]; returns [Hq, Lq, Dv] with causal offset S-Lq
    Hq, Lq, D = q.shape; Hkv, S, _ = K.shape; g = Hq // Hkv
    K = K.repeat_interleave(g, 0).float(); V = V.repeat_interleave(g, 0).float()
    s = torch.einsum('hqd,hkd->hqk', q.float(), K) * scale
    if cap > 0: s = cap * torch.tanh(s / cap)
    mask = torch.arange(S, device=q.device)[None, :] <= (torch.arange(Lq, device=q.device)[:, None] + S - Lq)
    s = s.masked_fill(~mask, float('-inf'))
    return torch.einsum('hqk,hkd->hqd', torch.softmax(s, -1), V)

def t_decode():
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        for (B, Hq, Hkv, D, S, splits, cap) in ((3, 32, 8, 128, 300, 1, 0.0), (3, 32, 8, 128, 1000, 4, 0.0), (2, 8, 2, 64, 77, 3, 30.0),
                                            (2, 12, 4, 80, 50, 2, 0.0), (1, 64, 8, 128, 513, 8, 0.0), (2, 4, 4, 128, 5, 1, 0.0), (2, 16, 1, 256, 100, 2, 0.0)):
            slots = B * S + 10
            kb = torch.randn(slots, Hkv, D, device=dev).to(dt); vb = torch.randn(slots, Hkv, D, device=dev).to(dt)
            q = torch.randn(B, Hq, D, device=dev).to(dt)
            lens = torch.randint(max(1, S // 2), S + 1, (B,), device=dev); lens[0] = S
            r2t = torch.zeros(B + 2, S + 4, dtype=torch.int32, device=dev)
            pool = torch.randperm(B + 2, device=dev)[:B]
            perm = torch.randperm(slots - 1, device=dev) + 1
            for b in range(B):
                r2t[pool[b], :lens[b]] = perm[b * S: b * S + int(lens[b])].int()
            o = torch.empty(B, Hq, D, device=dev, dtype=dt)
            al = torch.empty(B, Hq, splits, D, device=dev); ll = torch.empty(B, Hq, splits, device=dev)
            scale = 1 / math.sqrt(D)
            ops.decode_attention_paged(q, kb, vb, o, r2t, pool, lens, al, ll, splits, scale, cap)
            ref = torch.stack([attn_ref(q[b][:, None], kb[r2t[pool[b], :lens[b]].long()].transpose(0, 1), vb[r2t[pool[b], :lens[b]].long()].transpose(0, 1), scale, cap)[:, 0] for b in range(B)])
            e1 = (o.float() - ref).abs().max().item()
            indptr = torch.zeros(B + 1, dtype=torch.int32, device=dev); indptr[1:] = torch.cumsum(lens, 0)
            kvi = torch.empty(int(lens.sum()), dtype=torch.int32, device=dev)
            ops.create_kv_indices(r2t, pool, lens, indptr, None, kvi)
            o2 = torch.empty_like(o)
            ops.decode_attention_fwd(q, kb, vb, o2, indptr, kvi, al, ll, splits, splits, scale, cap)
            e2 = (o2.float() - ref).abs().max().item()
            report(f"decode {dt} B{B} Hq{Hq} Hkv{Hkv} D{D} S{S} splits{splits} cap{cap}", max(e1, e2), 2e-5 if dt == torch.float32 else 2e-2)

def t_extend():
    for dt in (torch.float32, torch.bfloat16):
        for (Hq, Hkv, D, pre, ext) in ((8, 2, 64, [0, 0, 0], [5, 17, 32]), (32, 8, 128, [40, 0, 130], [70, 129, 3]), (12, 4, 80, [10, 3], [33, 64]),
                                       (4, 4, 128, [100], [1]), (16, 1, 128, [7, 0], [20, 65]), (6, 2, 256, [20], [40])):
            B = len(pre); T = sum(ext); slots = sum(pre) + T + 5
            kb = torch.randn(slots, Hkv, D, device=dev).to(dt); vb = torch.randn(slots, Hkv, D, device=dev).to(dt)
            q = torch.randn(T, Hq, D, device=dev).to(dt); k = torch.randn(T, Hkv, D, device=dev).to(dt); v = torch.randn(T, Hkv, D, device=dev).to(dt)
            perm = (torch.randperm(slots - 1, device=dev) + 1).int()
            kvi = perm[:sum(pre)].contiguous() if sum(pre) else torch.zeros(1, dtype=torch.int32, device=dev)
            qo = torch.tensor([0] + list(torch.tensor(ext).cumsum(0)), dtype=torch.int32, device=dev)
            kvp = torch.tensor([0] + list(torch.tensor(pre).cumsum(0)), dtype=torch.int32, device=dev)
            o = torch.empty(T, Hq, D, device=dev, dtype=dt)
            scale = 1 / math.sqrt(D)
            ops.extend_attention_fwd(q, k, v, o, kb, vb, qo, kvp, kvi, None, True, None, max(ext), scale)
            err = 0.0
            for b in range(B):
                qs, qe = int(qo[b]), int(qo[b + 1]); ps, pe = int(kvp[b]), int(kvp[b + 1])
                K = torch.cat([kb[kvi[ps:pe].long()] if pe > ps else kb[:0], k[qs:qe]]).transpose(0, 1)
                V = torch.cat([vb[kvi[ps:pe].long()] if pe > ps else vb[:0], v[qs:qe]]).transpose(0, 1)
                ref = attn_ref(q[qs:qe].transpose(0, 1), K, V, scale).transpose(0, 1)
                err = max(err, (o[qs:qe].float() - ref).abs().max().item())
            report(f"extend {dt} Hq{Hq} Hkv{Hkv} D{D} pre{pre} ext{ext}", err, 2e-5 if dt == torch.float32 else 2e-2)

def t_linear():
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        for (M, N, K) in ((64, 6144, 4096), (1, 4096, 4096), (17, 1000, 512), (64, 4096, 14336), (33, 128256, 2048), (100, 512, 300), (5, 64, 8)):
            if dt == torch.float32 and N * K > 4096 * 4096: continue
            x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
            ref = x.double() @ w.double().t()
            y = ops.linear(x, w)
            tol = 1e-4 if dt == torch.float32 else 3e-2
            report(f"linear {dt} {M}x{N}x{K}", (y.double() - ref).abs().max().item(), tol)
    x = torch.randn(8, 512, device=dev).bfloat16(); w = torch.randn(1000, 512, device=dev).bfloat16() / 22
    y = ops.linear(x, w, out_dtype=torch.float32)
    report("linear out_f32 rounded", (y - (x.double() @ w.double().t()).float().bfloat16().float()).abs().max().item(), 3e-2)

def t_fp8():
    for (M, N, K) in ((64, 4096, 4096), (3, 1000, 1024), (16, 512, 14336), (100, 256, 512)):
        x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev) / math.sqrt(K)
        ws = (w.abs().amax(1, keepdim=True) / 448).float(); wq = (w / ws).clamp(-448, 448).to(torch.float8_e4m3fn)
        xq, xs = ops.per_token_quant_fp8(x)
        s_ref = x.float().abs().amax(1, keepdim=True).clamp(min=1e-10) / 448
        q_ref = (x.float() * (1.0 / s_ref)).clamp(-448, 448).to(torch.float8_e4m3fn)
        report(f"per_token_quant scale {M}x{K}", (xs - s_ref).abs().max().item(), 0)
        report(f"per_token_quant q {M}x{K}", (xq.float() - q_ref.float()).abs().max().item(), 0)
        y = ops.fp8_scaled_mm(xq, wq, xs, ws, torch.bfloat16)
        ref = ((xq.double() @ wq.double().t()) * xs.double() * ws.double().t())
        report(f"fp8_scaled_mm {M}x{N}x{K}", (y.double() - ref).abs().max().item(), 3e-2)

def t_sampling():
    for (B, V) in ((64, 128256), (3, 111), (5, 32000)):
        lg = torch.randn(B, V, device=dev)
        lg[0, 5] = lg[0, 77] = 100.0
        ids = ops.argmax(lg)
        report(f"argmax {B}x{V}", float((ids != torch.argmax(lg, -1)).sum().item()), 0)
        temp = torch.rand(B, device=dev) + 0.5
        pr = ops.softmax_temperature_(lg.clone(), temp)
        ref = torch.softmax(lg / temp[:, None], -1)
        report(f"softmax_temperature {B}x{V}", (pr - ref).abs().max().item(), 1e-6)
        ks = torch.randint(1, 50, (B,), device=dev, dtype=torch.int32); ps = torch.rand(B, device=dev) * 0.9 + 0.05
        mps = torch.zeros(B, device=dev)
        sp, si = ref.sort(-1, descending=True); cs = sp.cumsum(-1)
        keep_sorted = (torch.arange(V, device=dev)[None] < ks[:, None]) & ((cs - sp) <= ps[:, None])
        keep = torch.zeros_like(keep_sorted).scatter_(1, si, keep_sorted)
        bad = 0
        for trial in range(20):
            u = torch.rand(B, device=dev)
            tok = ops.top_k_top_p_min_p_sample(ref.contiguous(), ks, ps, mps, u)
            bad += int((~keep.gather(1, tok.long()[:, None])).sum().item())
        report(f"sample in-mask {B}x{V}", float(bad), 0)
        # distribution check on one row
    V = 1000; p = torch.softmax(torch.randn(1, V, device=dev) * 2, -1)
    ks = torch.tensor([20], dtype=torch.int32, device=dev); ps = torch.tensor([0.8], device=dev); mp = torch.tensor([0.0], device=dev)
    sp, si = p.sort(-1, descending=True); cs = sp.cumsum(-1)
    km = ((torch.arange(V, device=dev)[None] < 20) & ((cs - sp) <= 0.8))
    tgt = torch.zeros(V, device=dev); tgt[si[0][km[0]]] = sp[0][km[0]]; tgt /= tgt.sum()
    n = 20000; us = torch.rand(n, device=dev)
    toks = ops.top_k_top_p_min_p_sample(p.expand(n, V).contiguous(), ks.expand(n).contiguous(), ps.expand(n).contiguous(), mp.expand(n).contiguous(), us)
    emp = torch.bincount(toks.long(), minlength=V).float() / n
    report("sample distribution L1", (emp - tgt).abs().sum().item(), 0.05)

def t_embed():
    tab = torch.randn(1000, 512, device=dev).bfloat16(); ids = torch.randint(0, 1000, (77,), device=dev)
    report("embedding", (ops.embedding(ids, tab) - tab[ids]).abs().max().item(), 0)
    out = ops.embedding(ids, tab[250:500].contiguous(), 250, 500)
    ref = torch.where(((ids >= 250) & (ids < 500))[:, None], tab[ids], torch.zeros_like(tab[ids]))
    report("embedding shard", (out - ref).abs().max().item(), 0)

for f in (t_rmsnorm, t_silu, t_rope, t_index, t_embed, t_decode, t_extend, t_linear, t_fp8, t_sampling):
    t0 = time.time(); run(f); print(f"-- {f.__name__} {time.time() - t0:.1f}s", flush=True)
print("ALL OK" if all(results) else f"FAILURES: {results.count(False)}")
sys.exit(0 if all(results) else 1)
