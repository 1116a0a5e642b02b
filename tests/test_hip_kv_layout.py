"""The head-major KV cache (MI355_KV_HEAD_MAJOR): memory [heads, slots, d] seen through permute(1, 0, 2) as the
[slots, heads, d] tensor of MHATokenToKVPool.get_key_buffer (mem_cache/memory_pool.py:395-418), and its interleaved
form (MI355_KV_INTERLEAVED): keys and values as the halves of one [heads, slots, 2, d] allocation.  Every entry point
that reads or writes the multi-head cache must give the SAME bits over a head-major cache as over the
reference's layout holding the same values: the scatter, the RoPE launch that writes the cache, the prefill
projection that writes the cache, decode attention in every addressing mode and split plan (16-bit and e4m3fn
caches), the decode launch that starts from the qkv partials, and prefill attention over a cached prefix."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from sglang_custom_amd import ops as _ops
    return _ops


def head_major(buf):
    """A head-major cache holding the values of the contiguous [slots, heads, d] tensor `buf`."""
    s, h, d = buf.shape
    hm = torch.empty((h, s, d), dtype=buf.dtype, device=buf.device).permute(1, 0, 2)
    hm.copy_(buf)
    assert not hm.is_contiguous() or h == 1
    return hm


def relayout(kb, vb, layout):
    """(k, v) caches holding the values of the contiguous pair, head-major or as the halves of one
    [heads, slots, 2, d] allocation (head-major when the two head dims differ)."""
    if layout == "interleaved" and kb.shape == vb.shape:
        s, h, d = kb.shape
        kv = torch.empty((h, s, 2, d), dtype=kb.dtype, device=kb.device)
        k, v = kv[:, :, 0, :].permute(1, 0, 2), kv[:, :, 1, :].permute(1, 0, 2)
        k.copy_(kb)
        v.copy_(vb)
        return k, v
    return head_major(kb), head_major(vb)


LAYOUTS = pytest.mark.parametrize("layout", ["head_major", "interleaved"])


def bits(t):
    return t.contiguous().view(torch.uint8 if t.element_size() == 1 else torch.int16 if t.element_size() == 2 else torch.int32)


def paged_world(bs, maxlen, hkv, d, dtype, seed, kv8=False, dv=None):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, maxlen, (bs,), generator=g, dtype=torch.int64)
    lens[0] = maxlen
    slots = int(lens.sum()) + 8
    perm = (torch.randperm(slots - 1, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(bs + 2, maxlen + 8, dtype=torch.int32)
    rpi = torch.randperm(bs + 2, generator=g)[:bs].to(torch.int64)
    off = 0
    for b in range(bs):
        n = int(lens[b])
        r2t[rpi[b], :n] = perm[off:off + n]
        off += n
    kb = torch.randn(slots, hkv, d, generator=g).to(dtype).to(DEV)
    vb = torch.randn(slots, hkv, dv or d, generator=g).to(dtype).to(DEV)
    if kv8:
        kb, vb = kb.to(torch.float8_e4m3fn), vb.to(torch.float8_e4m3fn)
    return lens.to(DEV), r2t.to(DEV), rpi.to(DEV), kb, vb, slots


@LAYOUTS
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kv8", [False, True], ids=["cache16", "cache_fp8"])
@pytest.mark.parametrize("loc_dt", [torch.int64, torch.int32])
def test_scatter_into_a_head_major_cache(ops, dtype, kv8, loc_dt, layout):
    torch.manual_seed(0)
    for heads, kd, vd, tokens in [(8, 128, 128, 37), (1, 128, 128, 5), (3, 64, 40, 300), (2, 8, 16, 1)]:
        slots = 2 * tokens + 3
        cdt = torch.float8_e4m3fn if kv8 else dtype
        kb = torch.zeros(slots, heads, kd, dtype=cdt, device=DEV)
        vb = torch.zeros(slots, heads, vd, dtype=cdt, device=DEV)
        kh, vh = relayout(kb, vb, layout)
        rows = torch.randn(tokens, heads * (kd + vd) + 8, device=DEV, dtype=dtype) * 3
        if kv8:
            rows[0, :3] = torch.tensor([1000.0, -1000.0, 1e-3], dtype=dtype, device=DEV)
        k = rows[:, : heads * kd].view(tokens, heads, kd)          # row-strided views
        v = rows[:, heads * kd: heads * (kd + vd)].view(tokens, heads, vd)
        loc = (torch.randperm(slots - 1, device=DEV)[:tokens] + 1).to(loc_dt)
        ops.set_kv_buffer(kb, vb, loc, k, v)
        ops.set_kv_buffer(kh, vh, loc, k, v)
        torch.cuda.synchronize()
        assert torch.equal(bits(kh), bits(kb)) and torch.equal(bits(vh), bits(vb)), (heads, kd, vd)
        assert bits(kb).any()


def test_layouts_that_are_neither_are_refused(ops):
    from sglang_custom_amd._lib import Mi355Error

    dt = torch.bfloat16
    kb = torch.zeros(16, 4, 2 * 128, dtype=dt, device=DEV)[:, :, :128]     # a strided slice: neither layout
    k = torch.zeros(2, 4, 128, dtype=dt, device=DEV)
    loc = torch.tensor([1, 2], device=DEV)
    with pytest.raises(AssertionError):
        ops.set_kv_buffer(kb, kb, loc, k, k)
    # k head-major, v not
    good = torch.zeros(16, 4, 128, dtype=dt, device=DEV)
    with pytest.raises(AssertionError):
        ops.set_kv_buffer(head_major(good), good, loc, k, k)
    # an fp32 model: the head-aware scatter serves 16-bit rows only, and says so
    k32 = torch.zeros(2, 4, 128, dtype=torch.float32, device=DEV)
    g32 = head_major(torch.zeros(16, 4, 128, dtype=torch.float32, device=DEV))
    with pytest.raises(Mi355Error):
        ops.set_kv_buffer(g32, g32, loc, k32, k32)


@LAYOUTS
@pytest.mark.parametrize("dtype,neox,rot", [(torch.bfloat16, True, 128), (torch.float16, True, 128), (torch.bfloat16, False, 128),
                                            (torch.bfloat16, True, 64)])
def test_rope_launch_that_writes_a_head_major_cache(ops, dtype, neox, rot, layout):
    torch.manual_seed(1)
    D, hq, hkv, T = 128, 8, 4, 33
    slots = 80
    q = torch.randn(T, hq * D, device=DEV, dtype=dtype)
    k = torch.randn(T, hkv * D, device=DEV, dtype=dtype)
    v = torch.randn(T, hkv * D, device=DEV, dtype=dtype)
    cs = torch.cat([torch.randn(512, rot // 2).cos(), torch.randn(512, rot // 2).sin()], dim=1).to(DEV).contiguous()
    pos = torch.randint(0, 512, (T,), device=DEV, dtype=torch.int64)
    loc = (torch.randperm(slots - 1, device=DEV)[:T] + 1).to(torch.int64)
    kb = torch.randn(slots, hkv, D, device=DEV, dtype=dtype)
    vb = torch.randn(slots, hkv, D, device=DEV, dtype=dtype)
    kh, vh = relayout(kb, vb, layout)
    q1, k1 = q.clone(), k.clone()
    ops.rope_inplace(pos, q1, k1, D, cs, neox, v=v, k_buffer=kb, v_buffer=vb, loc=loc)
    q2, k2 = q.clone(), k.clone()
    ops.rope_inplace(pos, q2, k2, D, cs, neox, v=v, k_buffer=kh, v_buffer=vh, loc=loc)
    torch.cuda.synchronize()
    assert torch.equal(q1, q2) and torch.equal(k1, k2)
    assert torch.equal(bits(kh), bits(kb)) and torch.equal(bits(vh), bits(vb))


@LAYOUTS
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kv8", [False, True], ids=["cache16", "cache_fp8"])
@pytest.mark.parametrize("hq,hkv,d,bs,maxlen,splits,tickets", [
    (32, 8, 128, 9, 300, 1, False), (32, 8, 128, 9, 300, 4, True), (16, 2, 128, 6, 530, 3, False), (8, 8, 128, 5, 70, 2, True),
    (8, 1, 128, 19, 90, 1, False), (8, 4, 64, 7, 200, 2, False), (4, 2, 256, 3, 150, 1, False), (6, 3, 80, 4, 100, 2, False)])
def test_decode_attention_over_a_head_major_cache(ops, dtype, kv8, hq, hkv, d, bs, maxlen, splits, tickets, layout):
    lens, r2t, rpi, kb, vb, slots = paged_world(bs, maxlen, hkv, d, dtype, seed=hq + d + bs, kv8=kv8)
    kh, vh = relayout(kb, vb, layout)
    q = torch.randn(bs, hq, d, device=DEV, dtype=dtype)
    lg = torch.empty(bs * hq * splits * d, dtype=torch.float32, device=DEV)
    ls = torch.empty(bs * hq * splits, dtype=torch.float32, device=DEV)
    tk = torch.zeros(bs * hq, dtype=torch.int32, device=DEV) if tickets else None
    outs = []
    for k_, v_ in ((kb, vb), (kh, vh)):
        o = torch.full((bs, hq, d), float("nan"), device=DEV, dtype=dtype)
        ops.decode_attention_paged(q, k_, v_, o, r2t, rpi, lens, lg, ls, splits, d ** -0.5, tickets=tk)
        torch.cuda.synchronize()
        assert not torch.isnan(o.float()).any()
        outs.append(o)
    assert torch.equal(bits(outs[0]), bits(outs[1]))
    # the kv_indptr / kv_indices addressing and the entry point with its own split scratch
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32, device=DEV)
    kv_indptr[1:] = lens.cumsum(0).to(torch.int32)
    kv_indices = torch.cat([r2t[rpi[b], : int(lens[b])] for b in range(bs)]).contiguous()
    o2 = torch.empty_like(outs[0])
    ops.decode_attention_fwd(q, kh, vh, o2, kv_indptr, kv_indices, lg, ls, splits, 64, d ** -0.5, tickets=tk)
    torch.cuda.synchronize()
    assert torch.equal(bits(o2), bits(outs[0]))
    o3 = torch.empty_like(outs[0])
    ops.decode_attention_paged_static(q, kh, vh, o3, r2t, rpi, lens, splits, d ** -0.5)
    o4 = torch.empty_like(outs[0])
    ops.decode_attention_paged_static(q, kb, vb, o4, r2t, rpi, lens, splits, d ** -0.5)
    torch.cuda.synchronize()
    assert torch.equal(bits(o3), bits(o4))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_decode_attention_with_different_key_and_value_widths_over_a_head_major_cache(ops, dtype):
    """head_dim != v_head_dim: two head-major allocations with their own strides (the generic kernel)."""
    hq, hkv, d, dv, bs, maxlen = 6, 3, 80, 48, 5, 120
    lens, r2t, rpi, kb, vb, slots = paged_world(bs, maxlen, hkv, d, dtype, seed=11, dv=dv)
    q = torch.randn(bs, hq, d, device=DEV, dtype=dtype)
    outs = []
    for splits in (1, 3):
        lg = torch.empty(bs * hq * splits * dv, dtype=torch.float32, device=DEV)
        ls = torch.empty(bs * hq * splits, dtype=torch.float32, device=DEV)
        pair = []
        for k_, v_ in ((kb, vb), (head_major(kb), head_major(vb))):
            o = torch.full((bs, hq, dv), float("nan"), device=DEV, dtype=dtype)
            ops.decode_attention_paged(q, k_, v_, o, r2t, rpi, lens, lg, ls, splits, d ** -0.5)
            torch.cuda.synchronize()
            assert not torch.isnan(o.float()).any()
            pair.append(o)
        assert torch.equal(bits(pair[0]), bits(pair[1])), splits


@LAYOUTS
@pytest.mark.parametrize("dtype,cache_dtype", [(torch.bfloat16, torch.float32), (torch.float16, torch.float16)])
@pytest.mark.parametrize("kv8", [False, True], ids=["cache16", "cache_fp8"])
def test_decode_launch_from_qkv_partials_over_a_head_major_cache(ops, dtype, cache_dtype, kv8, layout):
    torch.manual_seed(2)
    D, hidden = 128, 1024
    for hq, hkv, bs, maxlen, splits in [(32, 8, 9, 300, 1), (16, 2, 6, 530, 3), (8, 1, 19, 90, 1)]:
        n = (hq + 2 * hkv) * D
        wp = ops.pack_weight(torch.randn(n, hidden, device=DEV, dtype=dtype) * 0.04)
        x = torch.randn(bs, hidden, device=DEV, dtype=dtype)
        lens, r2t, rpi, kb0, vb0, slots = paged_world(bs, maxlen, hkv, D, dtype, seed=hq + bs, kv8=kv8)
        loc = torch.stack([r2t[rpi[b], int(lens[b]) - 1] for b in range(bs)]).to(torch.int64)
        positions = lens - 1
        cs = torch.cat([torch.randn(maxlen + 8, D // 2).cos(), torch.randn(maxlen + 8, D // 2).sin()], dim=1).to(cache_dtype).to(DEV).contiguous()
        lg = torch.empty(bs * hq * 8 * D, dtype=torch.float32, device=DEV)
        ls = torch.empty(bs * hq * 8, dtype=torch.float32, device=DEV)
        tk = torch.zeros(bs * hq, dtype=torch.int32, device=DEV)
        part = ops.linear_packed_partials(x, wp, n)
        res = []
        for hm in (False, True):
            kb, vb = kb0.clone(), vb0.clone()
            if hm:
                kb, vb = relayout(kb, vb, layout)
            o = torch.empty(bs, hq, D, device=DEV, dtype=dtype)
            ops.decode_attention_fused_qkv(part, positions, loc, cs, kb, vb, o, r2t, rpi, lens, lg, ls, splits, D ** -0.5, tickets=tk)
            torch.cuda.synchronize()
            res.append((o, kb, vb))
        assert torch.equal(bits(res[0][0]), bits(res[1][0])), (hq, hkv)
        assert torch.equal(bits(res[0][1]), bits(res[1][1])) and torch.equal(bits(res[0][2]), bits(res[1][2]))


@LAYOUTS
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kv8", [False, True], ids=["cache16", "cache_fp8"])
@pytest.mark.parametrize("hq,hkv,d", [(8, 2, 128), (4, 4, 128), (6, 3, 64)])
def test_prefill_attention_over_a_cached_prefix_in_a_head_major_cache(ops, dtype, kv8, hq, hkv, d, layout):
    g = torch.Generator().manual_seed(hq * 7 + d)
    prefix = [0, 70, 300, 129]
    ext = [200, 64, 1, 130]
    B = len(ext)
    slots = sum(prefix) + 9
    kb = torch.randn(slots, hkv, d, generator=g).to(dtype).to(DEV)
    vb = torch.randn(slots, hkv, d, generator=g).to(dtype).to(DEV)
    if kv8:
        kb, vb = kb.to(torch.float8_e4m3fn), vb.to(torch.float8_e4m3fn)
    T = sum(ext)
    q = torch.randn(T, hq, d, generator=g).to(dtype).to(DEV)
    k = torch.randn(T, hkv, d, generator=g).to(dtype).to(DEV)
    v = torch.randn(T, hkv, d, generator=g).to(dtype).to(DEV)
    qo = torch.zeros(B + 1, dtype=torch.int32)
    qo[1:] = torch.tensor(ext).cumsum(0)
    kvp = torch.zeros(B + 1, dtype=torch.int32)
    kvp[1:] = torch.tensor(prefix).cumsum(0)
    kvi = (torch.randperm(slots - 1, generator=g)[: sum(prefix)] + 1).to(torch.int32)
    outs = []
    for k_, v_ in ((kb, vb), relayout(kb, vb, layout)):
        o = torch.full((T, hq, d), float("nan"), device=DEV, dtype=dtype)
        ops.extend_attention_fwd(q, k, v, o, k_, v_, qo.to(DEV), kvp.to(DEV), kvi.to(DEV), None, True, None, max(ext), d ** -0.5)
        torch.cuda.synchronize()
        assert not torch.isnan(o.float()).any()
        outs.append(o)
    assert torch.equal(bits(outs[0]), bits(outs[1]))


@LAYOUTS
@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "fp8_w8a8"])
def test_prefill_projection_that_writes_a_head_major_cache(ops, fp8, layout):
    torch.manual_seed(3)
    dtype = torch.bfloat16
    m, kdim, hq, hkv, D = 300, 512, 8, 2, 128
    n = (hq + 2 * hkv) * D
    x = torch.randn(m, kdim, device=DEV, dtype=dtype)
    w = torch.randn(n, kdim, device=DEV, dtype=dtype) * 0.05
    pos = torch.randint(0, 1000, (m,), device=DEV, dtype=torch.int64)
    cs = torch.cat([torch.randn(1000, D // 2).cos(), torch.randn(1000, D // 2).sin()], dim=1).to(DEV).contiguous()
    slots = 2 * m + 5
    loc = (torch.randperm(slots - 1, device=DEV)[:m] + 1).to(torch.int64)
    kw = {}
    if fp8:
        xq, xs = ops.per_token_quant_fp8(x)
        ws = (w.float().abs().amax(dim=1) / 448.0).clamp(min=1e-12)
        wq = (w.float() / ws[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
        x_in, w_in, kw = xq, wq, dict(x_scale=xs, w_scale=ws.contiguous(), out_dtype=dtype)
    else:
        x_in, w_in = x, w
    if not ops.qkv_proj_rope_tiled_supported(m, kdim, hq, hkv, D, D, True, dtype, fp8):
        pytest.skip("shape not served")
    res = []
    for hm in (False, True):
        kb = torch.zeros(slots, hkv, D, device=DEV, dtype=dtype)
        vb = torch.zeros(slots, hkv, D, device=DEV, dtype=dtype)
        if hm:
            kb, vb = relayout(kb, vb, layout)
        qkv = ops.qkv_proj_rope_tiled(x_in, w_in, hq, hkv, D, pos, cs, True, kb, vb, loc, **kw)
        torch.cuda.synchronize()
        res.append((qkv, kb, vb))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(bits(res[0][1]), bits(res[1][1])) and torch.equal(bits(res[0][2]), bits(res[1][2]))
    assert bits(res[0][1]).any()


@pytest.mark.parametrize("graphs", [False, True])
@pytest.mark.parametrize("quant,kvd", [(None, "auto"), ("w8a8_fp8", "fp8_e4m3")])
def test_model_runner_gives_the_same_logits_over_either_layout(graphs, quant, kvd):
    """Prefill, prefill over a cached prefix, eager and graph decode of a bf16 model with the pool head-major (what
    kv_layout="auto" picks on the GPU) against the reference's layout: logits equal bit for bit at every step
    and the caches hold the same values."""
    import cases
    from sglang_custom_amd import bench_one_batch as bob
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.managers.schedule_batch import Req
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
    from sglang_custom_amd.server_args import ServerArgs

    name, preset, dtn, lens, steps = [c for c in cases.E2E_CASES if c[2] == "bf16" and "d128" in c[1]][0]
    cfg = hf_config_dict(preset, "bfloat16")
    w = make_synthetic_llama_weights(cfg, seed=1234, dtype=torch.bfloat16)
    outs = []
    for layout in ("token_major", "auto"):
        sa = ServerArgs(dtype="bfloat16", max_total_tokens=2048, max_running_requests=16, disable_cuda_graph=not graphs,
                        cuda_graph_bs=[1, 2, 4], quantization=quant, kv_cache_dtype=kvd, kv_layout=layout)
        mr = ModelRunner(ModelConfig(cfg, dtype="bfloat16"), sa, weights={k: v.cuda() for k, v in w.items()})
        assert mr.token_to_kv_pool.head_major == (layout == "auto")
        if layout == "auto":      # keys | values of a (head, slot) side by side in one allocation
            kb0, vb0 = mr.token_to_kv_pool.get_kv_buffer(0)
            assert vb0.data_ptr() == kb0.data_ptr() + kb0.shape[2] * kb0.element_size()
        prompts = cases.e2e_prompts(cfg["vocab_size"], lens)
        bob.clear_pools(mr)
        reqs = [Req(i, p) for i, p in enumerate(prompts)]
        nxt, logits, batch = bob.extend(reqs, mr)
        got = [logits.clone()]
        for s in range(steps + 2):
            nxt, logits = bob.decode(nxt, batch, mr)
            got.append(logits.clone())
        kb, vb = mr.token_to_kv_pool.get_kv_buffer(0)
        outs.append((got, kb.contiguous().clone(), vb.contiguous().clone()))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(bits(outs[0][1]), bits(outs[1][1])) and torch.equal(bits(outs[0][2]), bits(outs[1][2]))
