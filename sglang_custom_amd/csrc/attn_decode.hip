// Decode attention (one new token per sequence) over the paged KV cache: split-KV flash
// decoding, stage 1 (partial softmax per KV split) + stage 2 (log-sum-exp merge).
//
// HBM-bound: every K and V row of the batch is read exactly once per (kv head); the G query
// heads that share a kv head are processed together so the row is reused from registers.
// K/V rows go straight from HBM to VGPRs with 16-byte loads: LPR = D*sizeof(T)/16 lanes
// cover one row, so one wave-instruction fetches 64/LPR whole rows (256-B bursts at D=128
// bf16).  Dot products are reduced across the LPR lanes of a row with xor-shuffles; every
// LPR-lane group keeps its own online-softmax state and the groups / waves are merged once
// at the end, so the key loop has no cross-group communication and no LDS traffic.
// No MFMA here on purpose: at G <= 8 the kernel sits on the HBM roof with VALU to spare.
#include "common.cuh"
#include <math.h>

namespace mi355 {

static constexpr float kLog2e = 1.4426950408889634f;
static constexpr float kLn2 = 0.6931471805599453f;

struct DecodeParams {
  const int32_t* page_table;
  const int64_t* req_pool_indices;
  const int64_t* seq_lens;
  const int32_t* kv_indptr;
  int64_t table_stride;
  int64_t q_stride;  // elements between consecutive sequences' rows of q
  float* part_o;    // [B, Hq, S, Dv]
  float* part_lse;  // [B, Hq, S]
  int mode, num_splits, hq, hkv;
  float sm_scale, logit_cap;
};

__device__ __forceinline__ void seq_range(const DecodeParams& p, int b, int64_t& page_off,
                                          int64_t& len) {
  if (p.mode == 0) {
    page_off = p.req_pool_indices[b] * p.table_stride;
    len = p.seq_lens[b];
  } else {
    page_off = p.kv_indptr[b];
    len = (int64_t)p.kv_indptr[b + 1] - page_off;
  }
}

__device__ __forceinline__ float safe_m(float m) { return m == -INFINITY ? 0.f : m; }

// grid: (hkv * (gtotal / G), num_splits, B); block 256 = 4 waves.
template <typename T, int D, int G, int U>
__global__ __launch_bounds__(256) void decode_attn_kernel(const T* __restrict__ q,
                                                          const T* __restrict__ kbuf,
                                                          const T* __restrict__ vbuf,
                                                          T* __restrict__ o, DecodeParams p) {
  constexpr int VE = V16<T>::n;   // elements per 16-B access
  constexpr int LPR = D / VE;     // lanes per K/V row
  constexpr int KPI = 64 / LPR;   // rows per wave-instruction
  static_assert(LPR >= 1 && LPR <= 64 && (LPR & (LPR - 1)) == 0, "head_dim not supported");
  __shared__ float sm_acc[4][G][D];
  __shared__ float sm_m[4][G];
  __shared__ float sm_l[4][G];

  const int gtotal = p.hq / p.hkv;
  const int groups_per_kv = gtotal / G;
  const int kvh = blockIdx.x / groups_per_kv;
  const int head0 = kvh * gtotal + (blockIdx.x % groups_per_kv) * G;
  const int split = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kslot = lane / LPR, dchunk = lane % LPR;

  int64_t page_off, len;
  seq_range(p, b, page_off, len);
  const int64_t chunk = (len + p.num_splits - 1) / p.num_splits;
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = k0 + chunk < len ? k0 + chunk : len;

  // q for this lane's d-chunk, pre-scaled.  Without a logit cap the softmax runs in the
  // log2 domain, so fold log2(e) in as well.
  const bool cap = p.logit_cap > 0.f;
  const float qmul = cap ? p.sm_scale : p.sm_scale * kLog2e;
  float qf[G][VE];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    Vec<T, VE> qv = ldvec<T, VE>(q + (int64_t)b * p.q_stride + (int64_t)(head0 + g) * D + dchunk * VE);
#pragma unroll
    for (int j = 0; j < VE; ++j) qf[g][j] = to_f32(qv.v[j]) * qmul;
  }

  float m[G], l[G], acc[G][VE];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = -INFINITY;
    l[g] = 0.f;
#pragma unroll
    for (int j = 0; j < VE; ++j) acc[g][j] = 0.f;
  }

  const int32_t* pt = p.page_table + page_off;
  const int64_t row_stride = (int64_t)p.hkv * D;
  const T* kbase = kbuf + (int64_t)kvh * D + dchunk * VE;
  const T* vbase = vbuf + (int64_t)kvh * D + dchunk * VE;
  constexpr int KEYS_PER_WAVE_ITER = KPI * U;

  for (int64_t base = k0 + (int64_t)wave * KEYS_PER_WAVE_ITER; base < k1;
       base += 4 * KEYS_PER_WAVE_ITER) {
    int64_t slot[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t key = base + u * KPI + kslot;
      valid[u] = key < k1;
      slot[u] = valid[u] ? (int64_t)pt[key] : 0;  // slot 0 always exists (padding sink row)
    }
    Vec<T, VE> kv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kv[u] = ldvec<T, VE>(kbase + slot[u] * row_stride);
#pragma unroll
    for (int u = 0; u < U; ++u) vv[u] = ldvec<T, VE>(vbase + slot[u] * row_stride);

    float s[U][G];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float kf[VE];
#pragma unroll
      for (int j = 0; j < VE; ++j) kf[j] = to_f32(kv[u].v[j]);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < VE; ++j) d = fmaf(qf[g][j], kf[j], d);
        s[u][g] = d;
      }
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int g = 0; g < G; ++g) s[u][g] += __shfl_xor(s[u][g], off, 64);

#pragma unroll
    for (int g = 0; g < G; ++g) {
      float mx = m[g];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float x = s[u][g];
        if (cap) x = p.logit_cap * tanhf(x / p.logit_cap) * kLog2e;
        x = valid[u] ? x : -INFINITY;
        s[u][g] = x;
        mx = fmaxf(mx, x);
      }
      const float ms = safe_m(mx);
      const float alpha = exp2f(m[g] - ms);
      m[g] = mx;
      float pu[U], psum = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        pu[u] = exp2f(s[u][g] - ms);
        psum += pu[u];
      }
      l[g] = l[g] * alpha + psum;
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        float a = acc[g][j] * alpha;
#pragma unroll
        for (int u = 0; u < U; ++u) a = fmaf(pu[u], to_f32(vv[u].v[j]), a);
        acc[g][j] = a;
      }
    }
  }

  // merge the KPI row-groups of this wave
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float mo = __shfl_xor(m[g], off, 64), lo = __shfl_xor(l[g], off, 64);
      const float mx = fmaxf(m[g], mo), ms = safe_m(mx);
      const float a = exp2f(m[g] - ms), bb = exp2f(mo - ms);
      l[g] = l[g] * a + lo * bb;
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        const float ao = __shfl_xor(acc[g][j], off, 64);
        acc[g][j] = acc[g][j] * a + ao * bb;
      }
      m[g] = mx;
    }
  }
  if (kslot == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int j = 0; j < VE; ++j) sm_acc[wave][g][dchunk * VE + j] = acc[g][j];
      if (dchunk == 0) {
        sm_m[wave][g] = m[g];
        sm_l[wave][g] = l[g];
      }
    }
  }
  __syncthreads();
  // merge the 4 waves and write out
  for (int idx = threadIdx.x; idx < G * D; idx += 256) {
    const int g = idx / D, d = idx - g * D;
    float mx = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mx = fmaxf(mx, sm_m[w][g]);
    const float ms = safe_m(mx);
    float lt = 0.f, at = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = exp2f(sm_m[w][g] - ms);
      lt += sm_l[w][g] * f;
      at += sm_acc[w][g][d] * f;
    }
    const float val = lt > 0.f ? at / lt : 0.f;
    const int head = head0 + g;
    if (p.num_splits == 1) {
      o[((int64_t)b * p.hq + head) * D + d] = from_f32<T>(val);
    } else {
      const int64_t ph = ((int64_t)b * p.hq + head) * p.num_splits + split;
      p.part_o[ph * D + d] = val;
      if (d == 0) p.part_lse[ph] = lt > 0.f ? mx * kLn2 + logf(lt) : -INFINITY;
    }
  }
}

// Any head_dim / v_head_dim (<= 1024): one wave per (head, split, sequence); lanes stride the
// feature axis.  Slow path for shapes outside the vectorised kernel (e.g. D = 80, 96, 576/512).
template <typename T>
__global__ __launch_bounds__(64) void decode_attn_generic_kernel(const T* __restrict__ q,
                                                                 const T* __restrict__ kbuf,
                                                                 const T* __restrict__ vbuf,
                                                                 T* __restrict__ o, DecodeParams p,
                                                                 int D, int Dv) {
  const int head = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x;
  const int kvh = head / (p.hq / p.hkv);
  int64_t page_off, len;
  seq_range(p, b, page_off, len);
  const int64_t chunk = (len + p.num_splits - 1) / p.num_splits;
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = k0 + chunk < len ? k0 + chunk : len;
  const T* qh = q + (int64_t)b * p.q_stride + (int64_t)head * D;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int64_t key = k0; key < k1; ++key) {
    const int64_t slot = p.page_table[page_off + key];
    const T* kr = kbuf + (slot * p.hkv + kvh) * (int64_t)D;
    const T* vr = vbuf + (slot * p.hkv + kvh) * (int64_t)Dv;
    float d = 0.f;
    for (int i = lane; i < D; i += 64) d = fmaf(to_f32(qh[i]), to_f32(kr[i]), d);
    d = wave_sum(d) * p.sm_scale;
    if (p.logit_cap > 0.f) d = p.logit_cap * tanhf(d / p.logit_cap);
    d *= kLog2e;
    const float mx = fmaxf(m, d);
    const float alpha = exp2f(m - mx), pe = exp2f(d - mx);
    l = l * alpha + pe;
    m = mx;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int dd = lane + 64 * i;
      if (dd < Dv) acc[i] = acc[i] * alpha + pe * to_f32(vr[dd]);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int dd = lane + 64 * i;
    if (dd < Dv) {
      const float val = l > 0.f ? acc[i] / l : 0.f;
      if (p.num_splits == 1) {
        o[((int64_t)b * p.hq + head) * Dv + dd] = from_f32<T>(val);
      } else {
        const int64_t ph = ((int64_t)b * p.hq + head) * p.num_splits + split;
        p.part_o[ph * Dv + dd] = val;
      }
    }
  }
  if (p.num_splits > 1 && lane == 0) {
    const int64_t ph = ((int64_t)b * p.hq + head) * p.num_splits + split;
    p.part_lse[ph] = l > 0.f ? m * kLn2 + logf(l) : -INFINITY;
  }
}

// Stage 2: o[b,h,:] = sum_s w_s * part_o[b,h,s,:], w_s = exp(lse_s - lse_total).
// grid (B*Hq), block 128.
template <typename T>
__global__ __launch_bounds__(128) void decode_merge_kernel(const float* __restrict__ part_o,
                                                           const float* __restrict__ part_lse,
                                                           T* __restrict__ o, int num_splits,
                                                           int Dv) {
  const int64_t bh = blockIdx.x;
  const float* lse = part_lse + bh * num_splits;
  float mx = -INFINITY;
  for (int s = 0; s < num_splits; ++s) mx = fmaxf(mx, lse[s]);
  const float ms = safe_m(mx);
  float tot = 0.f;
  for (int s = 0; s < num_splits; ++s) tot += expf(lse[s] - ms);
  for (int d = threadIdx.x; d < Dv; d += 128) {
    float a = 0.f;
    for (int s = 0; s < num_splits; ++s)
      a += expf(lse[s] - ms) * part_o[(bh * num_splits + s) * Dv + d];
    o[bh * Dv + d] = from_f32<T>(tot > 0.f ? a / tot : 0.f);
  }
}

template <typename T, int D, int U>
static bool launch_decode_d(const T* q, const T* kb, const T* vb, T* o, const DecodeParams& p,
                            int64_t bs, hipStream_t st) {
  const int gtotal = p.hq / p.hkv;
  const int G = gtotal % 8 == 0 ? 8 : gtotal % 4 == 0 ? 4 : gtotal % 2 == 0 ? 2 : 1;
  const dim3 grid((unsigned)(p.hkv * (gtotal / G)), (unsigned)p.num_splits, (unsigned)bs);
  const dim3 block(256);
  switch (G) {
    case 8: hipLaunchKernelGGL((decode_attn_kernel<T, D, 8, (U > 2 ? 2 : U)>), grid, block, 0, st, q, kb, vb, o, p); break;
    case 4: hipLaunchKernelGGL((decode_attn_kernel<T, D, 4, U>), grid, block, 0, st, q, kb, vb, o, p); break;
    case 2: hipLaunchKernelGGL((decode_attn_kernel<T, D, 2, U>), grid, block, 0, st, q, kb, vb, o, p); break;
    default: hipLaunchKernelGGL((decode_attn_kernel<T, D, 1, U>), grid, block, 0, st, q, kb, vb, o, p); break;
  }
  return true;
}

template <typename T>
static int launch_decode(const void* q_, const void* kb_, const void* vb_, void* o_,
                         const DecodeParams& p, int64_t bs, int D, int Dv, hipStream_t st) {
  const T* q = (const T*)q_;
  const T* kb = (const T*)kb_;
  const T* vb = (const T*)vb_;
  T* o = (T*)o_;
  const bool vec = D == Dv && aligned16(q) && aligned16(kb) && aligned16(vb) &&
                   p.q_stride % V16<T>::n == 0;
  constexpr int VE = V16<T>::n;
  if (vec && D == 64) launch_decode_d<T, 64, 4>(q, kb, vb, o, p, bs, st);
  else if (vec && D == 128) launch_decode_d<T, 128, 4>(q, kb, vb, o, p, bs, st);
  else if (vec && D == 256 && D / VE <= 64) launch_decode_d<T, 256, 2>(q, kb, vb, o, p, bs, st);
  else {
    if (D > 1024 || Dv > 1024) {
      set_error("decode_attention: head_dim %d / v_head_dim %d > 1024", D, Dv);
      return MI355_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((decode_attn_generic_kernel<T>), dim3((unsigned)p.hq, (unsigned)p.num_splits, (unsigned)bs),
                       dim3(64), 0, st, q, kb, vb, o, p, D, Dv);
  }
  if (p.num_splits > 1)
    hipLaunchKernelGGL((decode_merge_kernel<T>), dim3((unsigned)(bs * p.hq)), dim3(128), 0, st,
                       p.part_o, p.part_lse, o, p.num_splits, Dv);
  return MI355_OK;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_decode_num_splits(int64_t bs, int hkv, int64_t max_seq_len, int max_splits) {
  if (bs <= 0 || hkv <= 0 || max_splits <= 1) return 1;
  // Aim for >= 4 workgroups per CU (256 CUs) while keeping >= 128 keys per split.
  const int64_t wgs = bs * hkv;
  int64_t want = (4 * 256 + wgs - 1) / wgs;
  const int64_t by_len = max_seq_len / 128 > 0 ? max_seq_len / 128 : 1;
  if (want > by_len) want = by_len;
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  return (int)want;
}

int mi355_decode_attention(const void* q, const void* k_buffer, const void* v_buffer, void* o,
                           const int32_t* page_table, int mode, const int64_t* req_pool_indices,
                           const int64_t* seq_lens, const int32_t* kv_indptr,
                           int64_t table_stride, float* attn_logits, float* attn_lse,
                           int num_splits, int64_t bs, int hq, int hkv, int head_dim,
                           int v_head_dim, int64_t q_stride, int64_t num_slots, float sm_scale,
                           float logit_cap, int dtype, void* stream) {
  MI355_REQUIRE(q && k_buffer && v_buffer && o && page_table, MI355_ERR_INVALID,
                "decode_attention: null pointer");
  MI355_REQUIRE(mode == 0 || mode == 1, MI355_ERR_INVALID, "decode_attention: mode must be 0|1");
  if (mode == 0)
    MI355_REQUIRE(req_pool_indices && seq_lens && table_stride > 0, MI355_ERR_INVALID,
                  "decode_attention: mode 0 needs req_pool_indices, seq_lens, table_stride");
  else
    MI355_REQUIRE(kv_indptr, MI355_ERR_INVALID, "decode_attention: mode 1 needs kv_indptr");
  MI355_REQUIRE(bs >= 0 && hq > 0 && hkv > 0 && hq % hkv == 0 && head_dim > 0 && v_head_dim > 0 &&
                    num_slots > 0,
                MI355_ERR_INVALID, "decode_attention: bad shape hq=%d hkv=%d D=%d Dv=%d", hq, hkv,
                head_dim, v_head_dim);
  MI355_REQUIRE(q_stride >= (int64_t)hq * head_dim, MI355_ERR_INVALID,
                "decode_attention: q_stride smaller than a row");
  MI355_REQUIRE(num_splits >= 1 && num_splits <= 65535, MI355_ERR_INVALID,
                "decode_attention: num_splits %d out of range", num_splits);
  MI355_REQUIRE(num_splits == 1 || (attn_logits && attn_lse), MI355_ERR_INVALID,
                "decode_attention: split-KV needs attn_logits and attn_lse scratch");
  MI355_REQUIRE(bs <= 65535, MI355_ERR_UNSUPPORTED, "decode_attention: batch > 65535");
  if (bs == 0) return MI355_OK;
  DecodeParams p;
  p.page_table = page_table;
  p.req_pool_indices = req_pool_indices;
  p.seq_lens = seq_lens;
  p.kv_indptr = kv_indptr;
  p.table_stride = table_stride;
  p.q_stride = q_stride;
  p.part_o = attn_logits;
  p.part_lse = attn_lse;
  p.mode = mode;
  p.num_splits = num_splits;
  p.hq = hq;
  p.hkv = hkv;
  p.sm_scale = sm_scale;
  p.logit_cap = logit_cap;
  int rc = MI355_OK;
  MI355_DISPATCH_DTYPE(dtype, "decode_attention",
                       rc = launch_decode<T>(q, k_buffer, v_buffer, o, p, bs, head_dim, v_head_dim,
                                             (hipStream_t)stream));
  if (rc != MI355_OK) return rc;
  MI355_CHECK_LAUNCH("decode_attention");
  return MI355_OK;
}

}  // extern "C"
