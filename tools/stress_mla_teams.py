"""Dev tool: randomized soak of the two-teams-per-workgroup MLA decode kernels (16-bit and fp8 cache, with and without the
per-sequence split plan): random batch sizes, split counts, head counts and ragged lengths; every launch is compared with the
one-team kernel on the same inputs (development switch 10) and a few sequences with float64 attention."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import ops, _lib
_lib.use_dev_library(); lib = _lib.load()
lib.mi355_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
rank, rope = 512, 64
D = rank + rope
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = 0.0
for it in range(iters):
    bs = int(torch.randint(33, 260, (1,), generator=g))
    H = [16, 16, 20, 5, 32][int(torch.randint(0, 5, (1,), generator=g))]
    S = [2, 4, 6, 8, 16][int(torch.randint(0, 5, (1,), generator=g))]
    hi = [40, 300, 2000][int(torch.randint(0, 3, (1,), generator=g))]
    lens = torch.randint(1, hi + 1, (bs,), generator=g)
    total = int(lens.sum())
    fp8 = bool(torch.randint(0, 2, (1,), generator=g))
    plan = bool(torch.randint(0, 2, (1,), generator=g))
    rows = (torch.randn(total + 1, 1, D, generator=g) * 1.5).bfloat16().cuda()
    cache = rows.to(torch.float8_e4m3fn) if fp8 else rows
    r2t = torch.zeros(bs, int(lens.max()), dtype=torch.int32)
    perm = torch.randperm(total, generator=g) + 1
    pos = 0
    for b in range(bs):
        n = int(lens[b]); r2t[b, :n] = perm[pos:pos + n].int(); pos += n
    r2t = r2t.cuda(); rpi = torch.arange(bs, dtype=torch.int64, device='cuda'); lt = lens.cuda()
    q = torch.randn(bs, H, D, generator=g).bfloat16().cuda()
    lg = torch.empty(bs * H * S * rank, dtype=torch.float32, device='cuda'); ls = torch.empty(bs * H * S, dtype=torch.float32, device='cuda')
    tk = torch.zeros(bs * H, dtype=torch.int32, device='cuda')
    ck = torch.tensor([(int(lens.max()) + S - 1) // S], dtype=torch.int32, device='cuda') if plan else None
    outs = []
    for teams in (1, 0):
        lib.mi355_debug_set(10, teams)
        o = torch.full((bs, H, rank), float('nan'), dtype=torch.bfloat16, device='cuda')
        ops.decode_attention_mla(q, cache, o, r2t, rpi, lt, lg, ls, S, rank, D ** -0.5, tickets=tk, kv_chunk_keys=ck)
        torch.cuda.synchronize()
        assert not torch.isnan(o.float()).any() and not tk.any(), (it, teams)
        outs.append(o.float())
    d = (outs[0] - outs[1]).abs().max().item()
    wide = cache.float()
    for b in (0, bs // 2, bs - 1):
        kv = wide[r2t[b, :int(lens[b])].long(), 0].double()
        ref = torch.softmax((q[b].double() @ kv.t()) * D ** -0.5, -1) @ kv[:, :rank]
        e = (outs[1][b].double() - ref).abs().max().item() / max(ref.abs().max().item(), 1.0)
        worst = max(worst, e)
        assert e <= 2e-2, (it, b, e)
    assert d <= 4e-2, (it, d)
    print(f"iter {it}: bs={bs} H={H} S={S} max_len={int(lens.max())} fp8={fp8} plan={plan} grid={((H + 15) // 16) * S * bs}: |two - one| {d:.4f}", flush=True)
print("ok; worst error against float64:", worst)
