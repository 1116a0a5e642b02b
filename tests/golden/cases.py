"""Deterministic inputs for the golden vectors.  Imported by gen_golden.py (which feeds them to
the reference) and by the tests (which feed them to the oracle and to the HIP kernels), so the
fixtures only need to store OUTPUTS.  torch's CPU generator is bit-reproducible for a given
torch build; the build container and the GPU box run the same image."""
import numpy as np
import torch

DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def _g(seed):
    return torch.Generator().manual_seed(seed)


def randn(shape, seed, dtype, scale=1.0):
    return (scale * torch.randn(shape, generator=_g(seed), dtype=torch.float32)).to(dtype)


NORM_CASES = [(4, 512), (3, 111), (2, 4096), (5, 500)]          # (rows, hidden)
ACT_CASES = [(4, 128), (3, 1000), (2, 3584)]                   # (rows, d)
# (tokens, hq, hkv, head, rotary_dim, neox, llama3-scaling)
ROPE_CASES = [(9, 8, 2, 64, 64, True, False), (5, 4, 4, 128, 64, False, False),
              (7, 32, 8, 128, 128, True, True)]
LLAMA3 = dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
              original_max_position_embeddings=256)
# (B, Hq, Hkv, D, max_len)
DECODE_CASES = [(3, 8, 2, 64, 50), (2, 4, 4, 128, 70), (2, 32, 8, 128, 33), (1, 16, 1, 128, 129)]
# (Hq, Hkv, D, prefix_lens, extend_lens)
EXTEND_CASES = [(8, 2, 64, [0, 0, 0], [5, 17, 32]), (8, 2, 128, [40, 0, 13], [30, 70, 3]),
                (4, 4, 128, [20], [1]), (16, 1, 128, [7, 0], [20, 65])]
SAMPLING_CASES = [(4, 111), (3, 2048), (2, 8192)]              # (B, vocab)


def norm_inputs(i, dt):
    rows, h = NORM_CASES[i]
    return (randn((rows, h), 100 + i, dt), randn((rows, h), 200 + i, dt),
            (1 + 0.1 * torch.randn(h, generator=_g(300 + i), dtype=torch.float32)).to(dt))


def act_inputs(i, dt):
    rows, d = ACT_CASES[i]
    return randn((rows, 2 * d), 400 + i, dt)


def rope_inputs(i, dt):
    t, hq, hkv, hd, rd, neox, l3 = ROPE_CASES[i]
    pos = torch.randint(0, 2048, (t,), generator=_g(500 + i))
    return pos, randn((t, hq * hd), 510 + i, dt), randn((t, hkv * hd), 520 + i, dt)


def paged_inputs(seed, dt, B, Hkv, D, lens, extra_slots=7):
    """A scrambled paged KV cache: returns k_cache, v_cache, req_to_token, req_pool_indices."""
    total = int(sum(lens))
    slots = total + extra_slots
    g = _g(seed)
    k_cache = randn((slots + 1, Hkv, D), seed + 1, dt)
    v_cache = randn((slots + 1, Hkv, D), seed + 2, dt)
    perm = torch.randperm(slots, generator=g) + 1
    max_len = int(max(lens))
    req_pool_indices = torch.randperm(B + 3, generator=g)[:B].to(torch.int64)
    req_to_token = torch.zeros((B + 3, max_len + 4), dtype=torch.int32)
    p = 0
    for b in range(B):
        req_to_token[req_pool_indices[b], : lens[b]] = perm[p: p + lens[b]].to(torch.int32)
        p += lens[b]
    return k_cache, v_cache, req_to_token, req_pool_indices


def decode_inputs(i, dt):
    B, Hq, Hkv, D, max_len = DECODE_CASES[i]
    g = _g(600 + i)
    lens = torch.randint(max(1, max_len // 2), max_len + 1, (B,), generator=g)
    lens[0] = max_len
    k_cache, v_cache, r2t, pool = paged_inputs(610 + 10 * i, dt, B, Hkv, D, lens.tolist())
    q = randn((B, Hq, D), 605 + i, dt)
    return q, k_cache, v_cache, r2t, pool, lens.to(torch.int64)


def extend_inputs(i, dt):
    Hq, Hkv, D, pre, ext = EXTEND_CASES[i]
    B = len(pre)
    seq = [p + e for p, e in zip(pre, ext)]
    k_cache, v_cache, r2t, pool = paged_inputs(710 + 10 * i, dt, B, Hkv, D, seq)
    T = sum(ext)
    q = randn((T, Hq, D), 705 + i, dt)
    return (q, k_cache, v_cache, r2t, pool, torch.tensor(seq, dtype=torch.int64),
            torch.tensor(pre, dtype=torch.int64), torch.tensor(ext, dtype=torch.int64))


def sampling_inputs(i):
    B, V = SAMPLING_CASES[i]
    g = _g(800 + i)
    logits = 3.0 * torch.randn((B, V), generator=g, dtype=torch.float32)
    temps = torch.rand((B, 1), generator=g, dtype=torch.float32) + 0.5
    top_ks = torch.randint(1, 40, (B,), generator=g).to(torch.int32)
    top_ps = torch.rand((B,), generator=g, dtype=torch.float32) * 0.8 + 0.1
    min_ps = torch.rand((B,), generator=g, dtype=torch.float32) * 0.05
    return logits, temps, top_ks, top_ps, min_ps


# end-to-end: (name, preset, dtype, prompt lengths, decode steps)
E2E_CASES = [
    ("tiny64_f32", "tiny-llama-d64", "f32", (5, 17, 32), 8),
    ("tiny64_bf16", "tiny-llama-d64", "bf16", (5, 17, 32), 8),
    ("tiny128_f32", "tiny-llama-d128", "f32", (3, 40), 8),
    ("tiny128_bf16", "tiny-llama-d128", "bf16", (3, 40), 8),
]
E2E_PREFIX_CUT = 4   # second scenario: first cache prompt[:4], then extend with the rest


def e2e_prompts(preset_vocab, lens, seed=0):
    rng = np.random.RandomState(seed)
    return [rng.randint(0, preset_vocab, n).tolist() for n in lens]
