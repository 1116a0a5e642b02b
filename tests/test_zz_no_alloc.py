"""mi355_decode_attention_static -- the entry point that allocates nothing: its split-KV scratch is a
fixed device array inside the library.  Called through the C ABI (ctypes, raw pointers) and compared
with mi355_decode_attention given caller scratch: the two run the same kernels on the same inputs,
so the outputs must be bit-identical, and both must sit within the attention tolerance of the CPU
oracle.  Two calls in a row, the second with a smaller input, so that stale contents of the shared
array from the first call would show.  Collected last on purpose (file name)."""
import pytest
import torch

import cases
from oracle import ops_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF16 = 1


def _call(lib, static, q, kc, vc, r2t, pool, lens, splits, scratch=None):
    B, Hq, D = q.shape
    Hkv, Dv = vc.shape[1], vc.shape[2]
    o = torch.full((B, Hq, Dv), float("nan"), dtype=q.dtype, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    common_tail = (B, Hq, Hkv, D, Dv, q.stride(0), kc.shape[0], float(D ** -0.5), 0.0, BF16, st)
    if static:
        rc = lib.mi355_decode_attention_static(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), r2t.data_ptr(), 0,
                                               pool.data_ptr(), lens.data_ptr(), None, r2t.stride(0), splits, *common_tail)
    else:
        al, ll = scratch
        rc = lib.mi355_decode_attention(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), r2t.data_ptr(), 0,
                                        pool.data_ptr(), lens.data_ptr(), None, r2t.stride(0), al.data_ptr(), ll.data_ptr(),
                                        None, None, splits, *common_tail)
    assert rc == 0, lib.mi355_last_error()
    torch.cuda.synchronize()
    return o


def _inputs(seed, B, Hq, Hkv, D, lens):
    kc, vc, r2t, pool = cases.paged_inputs(seed, torch.bfloat16, B, Hkv, D, lens)
    q = cases.randn((B, Hq, D), seed + 5, torch.bfloat16)
    return q, kc, vc, r2t, pool, torch.tensor(lens, dtype=torch.int64)


def test_static_scratch_decode_attention_matches_the_allocating_entry_point():
    from sglang_custom_amd import _lib

    lib = _lib.load()
    # first call: 6 sequences x 32 heads x 8 splits; second: smaller batch, fewer heads, fewer
    # splits, shorter sequences -- its cells overlap the first call's in the shared array
    for (seed, B, Hq, Hkv, D, lens, splits) in ((900, 6, 32, 8, 128, [700, 333, 512, 64, 129, 1], 8),
                                                (910, 2, 8, 2, 128, [150, 47], 3)):
        cpu = _inputs(seed, B, Hq, Hkv, D, lens)
        q, kc, vc, r2t, pool, lt = (x.to(DEV) for x in cpu)
        al = torch.empty((B, Hq, splits, D), dtype=torch.float32, device=DEV)
        ll = torch.empty((B, Hq, splits), dtype=torch.float32, device=DEV)
        want = _call(lib, False, q, kc, vc, r2t, pool, lt, splits, (al, ll))
        got = _call(lib, True, q, kc, vc, r2t, pool, lt, splits)
        assert not torch.isnan(got.float()).any()
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (seed, "static != allocating entry point")
        ref = ops_ref.decode_attention(*cpu[:5], cpu[5], D ** -0.5, Hq != Hkv)
        err = (got.cpu().double() - ref.double()).abs().max().item()
        assert err <= 1.6e-2, (seed, err)   # one bf16 rounding of an output in [2, 4)


def test_static_scratch_rejects_sizes_beyond_the_built_in_array():
    from sglang_custom_amd import _lib

    lib = _lib.load()
    q, kc, vc, r2t, pool, lt = (x.to(DEV) for x in _inputs(920, 2, 8, 2, 128, [20, 9]))
    o = torch.empty_like(q)
    # 4096 x 64 x 8 x 129 x 4 B = 1 GiB of partials > 16 MiB: refused before anything is launched
    rc = lib.mi355_decode_attention_static(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), r2t.data_ptr(), 0,
                                           pool.data_ptr(), lt.data_ptr(), None, r2t.stride(0), 8, 4096, 64, 8, 128, 128,
                                           q.stride(0), kc.shape[0], 0.1, 0.0, BF16, None)
    assert rc == -2 and b"exceed the built-in" in lib.mi355_last_error()
    # one split needs no scratch at all: any batch is accepted (here the real, small one)
    rc = lib.mi355_decode_attention_static(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), r2t.data_ptr(), 0,
                                           pool.data_ptr(), lt.data_ptr(), None, r2t.stride(0), 1, 2, 8, 2, 128, 128,
                                           q.stride(0), kc.shape[0], 128 ** -0.5, 0.0, BF16, None)
    assert rc == 0, lib.mi355_last_error()
    torch.cuda.synchronize()
