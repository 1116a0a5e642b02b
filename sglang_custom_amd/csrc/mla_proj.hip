// The small steps between the projections of a multi-head-latent-attention decode layer, each as ONE launch:
//   mla_latent_norms  the output row of fused_qkv_a_proj_with_mqa [q_a | c_kv | k_pe] (or its split-K partials)
//                     -> q_a_layernorm(q_a), kv_a_layernorm(c_kv) and k_pe as three dense tensors;
//   mla_absorb        the per-head products that fold kv_b_proj into the query (q_nope . w_kc, with the rope part
//                     of q copied behind the result) and into the output (attn . w_vc, written in o_proj's row layout).
// Both are latency-bound at decode sizes (<= 64 rows, a few MB of L2-resident operands): what they buy is launches.
#include "common.cuh"

namespace mi355 {

// ------------------------------------------------------------------------------------------
// grid (rows, segments).  Segment 0 normalises columns [0, q_rank) with wq; segment 1 normalises
// [q_rank, q_rank + kv_rank) with wkv and copies the rope_dim columns behind them.  One 8-element vector
// per lane, the row held in registers: the arithmetic and its order are those of the one-chunk RMSNorm
// kernel, so the results are the bits of split + contiguous + rmsnorm.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mla_latent_norms_kernel(
    T* __restrict__ q_out, T* __restrict__ kv_out, T* __restrict__ pe_out, const T* __restrict__ x, int64_t x_stride,
    const float* __restrict__ part, int nsplit, int64_t split_stride, const float* __restrict__ part_xs,
    const float* __restrict__ part_ws, const T* __restrict__ wq, const T* __restrict__ wkv, int q_rank, int kv_rank,
    int rope, float eps, uint8_t* __restrict__ q8_out, float* __restrict__ q8_scale) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const int seg = (int)blockIdx.y + (q_rank ? 0 : 1);
  const int width = q_rank + kv_rank + rope;
  const int col0 = seg ? q_rank : 0;
  const int hidden = seg ? kv_rank : q_rank;
  const int nvec = hidden / 8;
  const int ncopy = seg ? rope / 8 : 0;
  const T* w = seg ? wkv : wq;
  T* orow = (seg ? kv_out : q_out) + row * hidden;
  const int t = threadIdx.x;
  const bool live = t < nvec + ncopy;
  const int col = col0 + 8 * (live ? t : 0);
  Vec<T, 8> wv = ldvec<T, 8>(w + 8 * (t < nvec ? t : 0));
  Vec<T, 8> a;
  if (part) {
    // split-K partials [split][rows][width] fp32: summed in split order and rounded once, as the GEMM's own combine
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = 0.f;
    const float* pr = part + row * width + col;
#pragma unroll 4
    for (int sp = 0; sp < nsplit; ++sp) {
#pragma unroll
      for (int j4 = 0; j4 < 8; j4 += 4) {
        const float4 f = *reinterpret_cast<const float4*>(pr + sp * split_stride + j4);
        sum[j4] += f.x; sum[j4 + 1] += f.y; sum[j4 + 2] += f.z; sum[j4 + 3] += f.w;
      }
    }
    if (part_xs) {   // partials of an fp8 GEMM: (acc * x_scale[m]) * w_scale[n]
      const float xs = part_xs[row];
#pragma unroll
      for (int j = 0; j < 8; ++j) a.v[j] = from_f32<T>((sum[j] * xs) * part_ws[col + j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) a.v[j] = from_f32<T>(sum[j] + 0.f);
    }
  } else {
    a = ldvec<T, 8>(x + row * x_stride + col);
  }
  if (t >= nvec && live) stvec<T, 8>(pe_out + row * rope + 8 * (t - nvec), a);
  float vals[8];
  float ss = 0.f;
  if (t < nvec) {
#pragma unroll
    for (int j = 0; j < 8; ++j) vals[j] = to_f32(a.v[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += vals[j] * vals[j];
  }
  ss = block_sum(ss, red);
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
  Vec<T, 8> o;
  float amax = 0.f;
  if (t < nvec) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o.v[j] = from_f32<T>((vals[j] * inv) * to_f32(wv.v[j]));
      amax = fmaxf(amax, fabsf(to_f32(o.v[j])));
    }
    stvec<T, 8>(orow + 8 * t, o);
  }
  if (q8_out && seg == 0) {
    // per-token fp8 quantisation of the normalised q_a row, for an fp8 q_b_proj (the arithmetic of
    // per_token_quant_fp8 on the stored 16-bit row)
    amax = wave_max(amax);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = fmaxf(amax, 1e-10f) / 448.0f;
    const float qinv = 1.0f / s;
    if (t == 0) q8_scale[row] = s;
    if (t < nvec) *reinterpret_cast<uint2*>(q8_out + row * hidden + 8 * t) = quant8_fp8<T>(o, qinv);
  }
}

// ------------------------------------------------------------------------------------------
// out[m, h, n] = sum_k a[m, h, k] * w_t[h, n, k]   (m < M <= 16 MT, k < K, n < N), 16-bit in, fp32 accumulate.
// One workgroup per (16 output features, head): the weight rows are the MFMA's A operand and the activations
// its B operand, so a lane ends with four consecutive features of one row -- one 8-byte store.  Operands come
// straight from L2 (16-byte fragments).  A long reduction is cut over the KW waves of the workgroup (each wave
// has all its loads in flight at once instead of walking K in dependent rounds); the partial tiles meet in
// LDS and are summed in wave order, wave w finishing row tiles w, w + KW, ...  Workgroup x == N / 16
// (present when tail > 0) copies a[m, h, K .. K + tail) behind the N outputs of each (m, h).
// ------------------------------------------------------------------------------------------
template <typename T, int MT, int KW>
__global__ __launch_bounds__(64 * KW) void mla_absorb_kernel(T* __restrict__ out, const T* __restrict__ a,
                                                             const T* __restrict__ w_t, int M, int N, int K, int tail,
                                                             int64_t a_row_stride, int64_t a_head_stride,
                                                             int64_t out_row_stride, int64_t out_head_stride) {
  using frag = typename Mfma<T>::frag;
  __shared__ f32x4 red[KW > 1 ? KW * MT * 64 : 1];
  const int h = blockIdx.y, nt = blockIdx.x, l = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* ah = a + (int64_t)h * a_head_stride;
  T* oh = out + (int64_t)h * out_head_stride;
  if (nt * 16 >= N) {
    const int per = tail / 8;
    for (int idx = threadIdx.x; idx < M * per; idx += 64 * KW) {
      const int m = idx / per, c = idx - m * per;
      stvec<T, 8>(oh + m * out_row_stride + N + 8 * c, ldvec<T, 8>(ah + m * a_row_stride + K + 8 * c));
    }
    return;
  }
  const int i = l & 15, g = l >> 4;
  const int kw = K / KW, kbeg = wave * kw;
  const T* wrow = w_t + ((int64_t)h * N + nt * 16 + i) * K + kbeg + 8 * g;
  const T* arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + i;
    arow[mt] = ah + (int64_t)(m < M ? m : M - 1) * a_row_stride + kbeg + 8 * g;   // rows past M: computed, never stored
  }
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k0 = 0; k0 < kw; k0 += 32) {
    const frag wf = *reinterpret_cast<const frag*>(wrow + k0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      acc[mt] = Mfma<T>::run(wf, *reinterpret_cast<const frag*>(arow[mt] + k0), acc[mt]);
  }
  auto store_tile = [&](int mt, const f32x4& v) {
    const int m = mt * 16 + i;
    if (m < M) {
      Vec<T, 4> o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o.v[r] = from_f32<T>(v[r]);
      stvec<T, 4>(oh + m * out_row_stride + nt * 16 + 4 * g, o);
    }
  };
  if constexpr (KW == 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) store_tile(mt, acc[mt]);
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[(wave * MT + mt) * 64 + l] = acc[mt];
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (mt % KW != wave) continue;
      f32x4 v = red[mt * 64 + l];
#pragma unroll
      for (int w = 1; w < KW; ++w) v += red[(w * MT + mt) * 64 + l];
      store_tile(mt, v);
    }
  }
}

template <typename T>
static void launch_absorb(void* out, const void* a, const void* w_t, int m, int heads, int n, int k, int tail,
                          int64_t ars, int64_t ahs, int64_t ors, int64_t ohs, hipStream_t st) {
  const dim3 grid((unsigned)(n / 16 + (tail ? 1 : 0)), (unsigned)heads);
  // every wave of a workgroup reduces at least 128 features (four MFMA steps in flight together)
  const int kw = k % 512 == 0 ? 4 : (k % 256 == 0 ? 2 : 1);
#define LAUNCH(MT_, KW_)                                                                                        \
  hipLaunchKernelGGL((mla_absorb_kernel<T, MT_, KW_>), grid, dim3(64 * KW_), 0, st, (T*)out, (const T*)a,       \
                     (const T*)w_t, m, n, k, tail, ars, ahs, ors, ohs)
#define LAUNCH_MT(KW_)             \
  do {                             \
    if (m <= 16) LAUNCH(1, KW_);   \
    else if (m <= 32) LAUNCH(2, KW_); \
    else if (m <= 48) LAUNCH(3, KW_); \
    else LAUNCH(4, KW_);           \
  } while (0)
  if (kw == 4) LAUNCH_MT(4);
  else if (kw == 2) LAUNCH_MT(2);
  else LAUNCH_MT(1);
#undef LAUNCH_MT
#undef LAUNCH
}

// ------------------------------------------------------------------------------------------
// The glue of MLA's multi-head prefill (deepseek_v2.py:1131-1179) as one launch, one workgroup per token:
// rotate q's rope part in place and the token's k_pe (interleaved or neox pairs, the rounding of rope_kernel),
// assemble k [T, H, nope + rope] = [kv's nope part | roped k_pe] and v [T, H, vd] from kv_b_proj's output
// kv [T, H, nope + vd], and write the latent row [normed c_kv | roped k_pe] to the cache slot of the token
// (16-bit, or e4m3fn with the scatter kernel's arithmetic).  All widths multiples of 8, rope <= 256.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mla_prefill_assemble_kernel(
    T* __restrict__ q, T* __restrict__ k, T* __restrict__ v, const T* __restrict__ kv, const T* __restrict__ k_pe,
    int64_t k_pe_stride, const T* __restrict__ c_kv, int64_t c_kv_stride, const int64_t* __restrict__ positions,
    const void* __restrict__ cos_sin, int cache_f32, int is_neox, void* __restrict__ cache,
    const int64_t* __restrict__ out_loc, int cache_fp8, int H, int nope, int rope, int vd, int rank, int64_t q_stride) {
  __shared__ __attribute__((aligned(16))) T s_pe[256];
  const int64_t t = blockIdx.x;
  const int64_t pos = positions[t];
  const int half = rope / 2;
  auto cs = [&](int idx) -> float {
    return cache_f32 ? static_cast<const float*>(cos_sin)[pos * rope + idx]
                     : to_f32(static_cast<const T*>(cos_sin)[pos * rope + idx]);
  };
  // pair i of a rope vector x: (x[2 i], x[2 i + 1]) interleaved, (x[i], x[i + half]) neox
  const int ia = is_neox ? 1 : 2, ib = is_neox ? half : 1;
  const T* pe = k_pe + t * k_pe_stride;
  for (int i = threadIdx.x; i < half; i += 256) {
    T o1, o2;
    rope_pair<T>(to_f32(pe[ia * i]), to_f32(pe[ia * i + ib]), cs(i), cs(half + i), o1, o2);
    s_pe[ia * i] = o1;
    s_pe[ia * i + ib] = o2;
  }
  T* qt = q + t * q_stride;
  for (int it = threadIdx.x; it < H * half; it += 256) {
    const int h = it / half, i = it - h * half;
    T* qp = qt + (int64_t)h * (nope + rope) + nope;
    T o1, o2;
    rope_pair<T>(to_f32(qp[ia * i]), to_f32(qp[ia * i + ib]), cs(i), cs(half + i), o1, o2);
    qp[ia * i] = o1;
    qp[ia * i + ib] = o2;
  }
  __syncthreads();
  // copies in 16-byte vectors: per head nope / 8 + rope / 8 vectors of k and vd / 8 of v
  const int vn = nope / 8, vr = rope / 8, vv = vd / 8, per_head = vn + vr + vv;
  const T* kvt = kv + t * (int64_t)H * (nope + vd);
  T* kt = k + t * (int64_t)H * (nope + rope);
  T* vt = v + t * (int64_t)H * vd;
  for (int it = threadIdx.x; it < H * per_head; it += 256) {
    const int h = it / per_head, c = it - h * per_head;
    if (c < vn) stvec<T, 8>(kt + (int64_t)h * (nope + rope) + 8 * c, ldvec<T, 8>(kvt + (int64_t)h * (nope + vd) + 8 * c));
    else if (c < vn + vr) stvec<T, 8>(kt + (int64_t)h * (nope + rope) + nope + 8 * (c - vn), *reinterpret_cast<const Vec<T, 8>*>(s_pe + 8 * (c - vn)));
    else stvec<T, 8>(vt + (int64_t)h * vd + 8 * (c - vn - vr), ldvec<T, 8>(kvt + (int64_t)h * (nope + vd) + nope + 8 * (c - vn - vr)));
  }
  // the latent cache row
  const int64_t slot = out_loc[t];
  const int row_vecs = (rank + rope) / 8;
  for (int c = threadIdx.x; c < row_vecs; c += 256) {
    const Vec<T, 8> x = c < rank / 8 ? ldvec<T, 8>(c_kv + t * c_kv_stride + 8 * c)
                                     : *reinterpret_cast<const Vec<T, 8>*>(s_pe + 8 * (c - rank / 8));
    if (cache_fp8)
      *reinterpret_cast<uint2*>(static_cast<uint8_t*>(cache) + slot * (rank + rope) + 8 * c) = quant8_fp8<T>(x, 1.0f);
    else
      stvec<T, 8>(static_cast<T*>(cache) + slot * (rank + rope) + 8 * c, x);
  }
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_mla_latent_norms(void* q_out, void* kv_out, void* pe_out, const void* x, int64_t x_row_stride,
                           const float* part, int nsplit, const float* part_x_scale, const float* part_w_scale,
                           const void* q_weight, const void* kv_weight, uint8_t* q_fp8_out, float* q_fp8_scale,
                           int64_t rows, int q_rank, int kv_rank, int rope_dim, float eps, int dtype, void* stream) {
  MI355_REQUIRE(kv_out && pe_out && kv_weight && (x || part) && !(x && part), MI355_ERR_INVALID,
                "mla_latent_norms: null pointer (exactly one of x and part is given)");
  MI355_REQUIRE((q_rank == 0) == (q_out == nullptr) && (q_rank == 0) == (q_weight == nullptr), MI355_ERR_INVALID,
                "mla_latent_norms: q_out and q_weight go with q_rank > 0");
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "mla_latent_norms: only 16-bit dtypes");
  MI355_REQUIRE((q_fp8_out == nullptr) == (q_fp8_scale == nullptr) && (!q_fp8_out || q_rank > 0) &&
                    (reinterpret_cast<uintptr_t>(q_fp8_out) & 7) == 0,
                MI355_ERR_INVALID, "mla_latent_norms: q_fp8_out and q_fp8_scale come together, 8-byte aligned, with q_rank > 0");
  MI355_REQUIRE(rows >= 0 && q_rank >= 0 && kv_rank > 0 && rope_dim > 0 && q_rank % 8 == 0 && kv_rank % 8 == 0 &&
                    rope_dim % 8 == 0 && q_rank <= 2048 && kv_rank + rope_dim <= 2048,
                MI355_ERR_UNSUPPORTED,
                "mla_latent_norms: needs q_rank, kv_rank, rope_dim %% 8 == 0, q_rank <= 2048, kv_rank + rope_dim <= 2048 "
                "(q_rank=%d kv_rank=%d rope_dim=%d)", q_rank, kv_rank, rope_dim);
  const int64_t width = (int64_t)q_rank + kv_rank + rope_dim;
  MI355_REQUIRE(part || (x_row_stride >= width && x_row_stride % 8 == 0 && aligned16(x)), MI355_ERR_INVALID,
                "mla_latent_norms: x needs a 16-byte aligned base and a row stride %% 8 == 0 of at least %lld",
                (long long)width);
  MI355_REQUIRE(!part || (nsplit >= 1 && nsplit <= 64 && aligned16(part)), MI355_ERR_INVALID,
                "mla_latent_norms: bad partials (nsplit=%d)", nsplit);
  MI355_REQUIRE((part_x_scale == nullptr) == (part_w_scale == nullptr) && (!part_x_scale || part), MI355_ERR_INVALID,
                "mla_latent_norms: scales come in pairs and only with partials");
  MI355_REQUIRE(aligned16(kv_out) && aligned16(pe_out) && aligned16(kv_weight) && (!q_out || (aligned16(q_out) && aligned16(q_weight))),
                MI355_ERR_INVALID, "mla_latent_norms: misaligned pointer");
  if (rows == 0) return MI355_OK;
  const dim3 grid((unsigned)rows, q_rank ? 2 : 1), block(256);
#define LAUNCH(T)                                                                                                   \
  hipLaunchKernelGGL((mla_latent_norms_kernel<T>), grid, block, 0, (hipStream_t)stream, (T*)q_out, (T*)kv_out,      \
                     (T*)pe_out, (const T*)x, x_row_stride, part, nsplit, rows * width, part_x_scale, part_w_scale, \
                     (const T*)q_weight, (const T*)kv_weight, q_rank, kv_rank, rope_dim, eps, q_fp8_out, q_fp8_scale)
  if (dtype == MI355_BF16) LAUNCH(bf16_t);
  else LAUNCH(f16_t);
#undef LAUNCH
  MI355_CHECK_LAUNCH("mla_latent_norms");
  return MI355_OK;
}

int mi355_mla_absorb(void* out, const void* a, const void* w_t, int64_t m, int heads, int n, int k, int tail,
                     int64_t a_row_stride, int64_t a_head_stride, int64_t out_row_stride, int64_t out_head_stride,
                     int dtype, void* stream) {
  MI355_REQUIRE(out && a && w_t, MI355_ERR_INVALID, "mla_absorb: null pointer");
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "mla_absorb: only 16-bit dtypes");
  MI355_REQUIRE(m >= 0 && m <= 64 && heads >= 1 && heads <= 65535 && n >= 16 && n % 16 == 0 && k >= 32 && k % 32 == 0 &&
                    tail >= 0 && tail % 8 == 0,
                MI355_ERR_UNSUPPORTED,
                "mla_absorb: needs m <= 64, n %% 16 == 0, k %% 32 == 0, tail %% 8 == 0 (m=%lld n=%d k=%d tail=%d)",
                (long long)m, n, k, tail);
  MI355_REQUIRE(a_head_stride >= k + tail && out_head_stride >= n + tail && a_row_stride >= a_head_stride &&
                    out_row_stride >= out_head_stride && a_row_stride % 8 == 0 && a_head_stride % 8 == 0 &&
                    out_row_stride % 8 == 0 && out_head_stride % 8 == 0,
                MI355_ERR_INVALID, "mla_absorb: strides must be multiples of 8 elements and cover a head's features");
  MI355_REQUIRE(aligned16(out) && aligned16(a) && aligned16(w_t), MI355_ERR_INVALID, "mla_absorb: misaligned pointer");
  if (m == 0) return MI355_OK;
  if (dtype == MI355_BF16)
    launch_absorb<bf16_t>(out, a, w_t, (int)m, heads, n, k, tail, a_row_stride, a_head_stride, out_row_stride,
                          out_head_stride, (hipStream_t)stream);
  else
    launch_absorb<f16_t>(out, a, w_t, (int)m, heads, n, k, tail, a_row_stride, a_head_stride, out_row_stride,
                         out_head_stride, (hipStream_t)stream);
  MI355_CHECK_LAUNCH("mla_absorb");
  return MI355_OK;
}

int mi355_mla_prefill_assemble(void* q, void* k, void* v, const void* kv, const void* k_pe, int64_t k_pe_stride,
                               const void* c_kv, int64_t c_kv_stride, const int64_t* positions, const void* cos_sin_cache,
                               int cache_dtype, int is_neox, void* kv_cache, const int64_t* out_cache_loc, int64_t tokens,
                               int heads, int nope_dim, int rope_dim, int v_dim, int kv_lora_rank, int64_t q_stride,
                               int64_t num_slots, int dtype, void* stream) {
  MI355_REQUIRE(q && k && v && kv && k_pe && c_kv && positions && cos_sin_cache && kv_cache && out_cache_loc,
                MI355_ERR_INVALID, "mla_prefill_assemble: null pointer");
  const bool kv8 = (dtype & MI355_KV_FP8) != 0;
  dtype &= 0xff;
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "mla_prefill_assemble: only 16-bit dtypes");
  MI355_REQUIRE(cache_dtype == MI355_F32 || cache_dtype == dtype, MI355_ERR_UNSUPPORTED,
                "mla_prefill_assemble: cos_sin_cache must be fp32 or the tensor dtype");
  MI355_REQUIRE(tokens >= 0 && heads >= 1 && nope_dim > 0 && rope_dim > 0 && v_dim > 0 && kv_lora_rank > 0 && num_slots > 0 &&
                    nope_dim % 8 == 0 && rope_dim % 8 == 0 && rope_dim <= 256 && v_dim % 8 == 0 && kv_lora_rank % 8 == 0,
                MI355_ERR_UNSUPPORTED,
                "mla_prefill_assemble: widths must be multiples of 8 and rope_dim <= 256 (nope=%d rope=%d v=%d rank=%d)",
                nope_dim, rope_dim, v_dim, kv_lora_rank);
  MI355_REQUIRE(q_stride >= (int64_t)heads * (nope_dim + rope_dim) && q_stride % 8 == 0 && k_pe_stride >= rope_dim &&
                    k_pe_stride % 8 == 0 && c_kv_stride >= kv_lora_rank && c_kv_stride % 8 == 0,
                MI355_ERR_INVALID, "mla_prefill_assemble: row strides must cover the rows and be multiples of 8");
  MI355_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(kv) && aligned16(k_pe) && aligned16(c_kv) &&
                    (reinterpret_cast<uintptr_t>(kv_cache) & (kv8 ? 7 : 15)) == 0,
                MI355_ERR_INVALID, "mla_prefill_assemble: misaligned pointer");
  if (tokens == 0) return MI355_OK;
#define LAUNCH(T)                                                                                                    \
  hipLaunchKernelGGL((mla_prefill_assemble_kernel<T>), dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream,    \
                     (T*)q, (T*)k, (T*)v, (const T*)kv, (const T*)k_pe, k_pe_stride, (const T*)c_kv, c_kv_stride,    \
                     positions, cos_sin_cache, cache_dtype == MI355_F32 ? 1 : 0, is_neox, kv_cache, out_cache_loc,   \
                     kv8 ? 1 : 0, heads, nope_dim, rope_dim, v_dim, kv_lora_rank, q_stride)
  if (dtype == MI355_BF16) LAUNCH(bf16_t);
  else LAUNCH(f16_t);
#undef LAUNCH
  MI355_CHECK_LAUNCH("mla_prefill_assemble");
  return MI355_OK;
}

}  // extern "C"
