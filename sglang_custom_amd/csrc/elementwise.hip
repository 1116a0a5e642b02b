// Bandwidth-bound row kernels: RMSNorm, fused add+RMSNorm, SiLU-and-mul, RoPE (+KV write),
// KV-cache scatter, embedding gather.  All use 16-byte vector accesses when the row length
// and pointers allow it and fall back to scalar accesses otherwise.
#include "common.cuh"
#include <algorithm>

namespace mi355 {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

// Pending L2 warm-up (mi355_prefetch_next): taken by the next launch that can carry it.
static thread_local WeightPrefetch g_pf;
static thread_local bool g_pf_pending = false;
static int take_prefetch(WeightPrefetch* d) {
  d->pf_steps = 0;
  if (!g_pf_pending) return 0;
  g_pf_pending = false;
  *d = g_pf;
  return d->nblocks * d->splits;
}

// ------------------------------------------------------------------------------------------
// RMSNorm.  One workgroup per row.  The row is held in registers between the sum-of-squares
// pass and the scale pass when it fits (CHUNKS vectors per thread), so HBM sees each element
// once; longer rows are re-read (from L2).
// ------------------------------------------------------------------------------------------
template <typename T, bool ADD, int CHUNKS, int BT = 256>
__global__ __launch_bounds__(BT) void rmsnorm_vec_kernel(T* __restrict__ out, T* __restrict__ x,
                                                          T* __restrict__ residual,
                                                          const T* __restrict__ w, int64_t hidden,
                                                          int64_t x_stride, int64_t out_stride,
                                                          float eps, WeightPrefetch pf, int pf_wgs,
                                                          const float* __restrict__ part, int nsplit,
                                                          const float* __restrict__ part_xs,
                                                          const float* __restrict__ part_ws,
                                                          uint8_t* __restrict__ q_out,
                                                          float* __restrict__ q_scale) {
  constexpr int VE = V16<T>::n;
  __shared__ float red[16];
  if ((int)blockIdx.x < pf_wgs) {   // warm-up riders come first so they keep the GEMM's XCD mapping
    prefetch_weights(pf, blockIdx.x);
    return;
  }
  const int64_t row = blockIdx.x - pf_wgs;
  T* xr = x + row * x_stride;
  T* rr = ADD ? residual + row * hidden : nullptr;
  T* orow = out + row * out_stride;
  const int nvec = (int)(hidden / VE);
  float vals[CHUNKS][VE];
  float ss = 0.f;
  // the norm weight is fetched up front so that its latency overlaps the row loads and the reduction
  Vec<T, VE> wv[CHUNKS];
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int vi = threadIdx.x + c * BT;
    wv[c] = ldvec<T, VE>(w + (int64_t)(vi < nvec ? vi : 0) * VE);
  }
  // Phase 1: every load of the row is issued before anything is consumed or stored (a store to the
  // residual between two chunks would otherwise pin the second chunk's loads behind it): the launch
  // pays one memory round trip, not one per chunk.  Split-K partials are summed split-major so that
  // the loads of all chunks of a split are in flight together; the order per element stays split 0, 1, ...
  Vec<T, VE> a[CHUNKS], b[CHUNKS];
  float sum[CHUNKS][VE];
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int vi = threadIdx.x + c * BT;
    const int64_t vo = (int64_t)(vi < nvec ? vi : 0) * VE;
    if (ADD) b[c] = ldvec<T, VE>(rr + vo);
    if (!part) a[c] = ldvec<T, VE>(xr + vo);
#pragma unroll
    for (int j = 0; j < VE; ++j) sum[c][j] = 0.f;
  }
  if (part) {
    // x is still the split-K partials of the GEMM that produced it ([split][rows][hidden] fp32):
    // sum in split order and round once, exactly as that GEMM's own combine would have.
    const int64_t split_stride = (int64_t)(gridDim.x - pf_wgs) * hidden;
    const float* pr = part + row * hidden;
#pragma unroll 4
    for (int sp = 0; sp < nsplit; ++sp) {
#pragma unroll
      for (int c = 0; c < CHUNKS; ++c) {
        const int vi = threadIdx.x + c * BT;
        const int64_t vo = (int64_t)(vi < nvec ? vi : 0) * VE;
#pragma unroll
        for (int j4 = 0; j4 < VE; j4 += 4) {
          const float4 f = *reinterpret_cast<const float4*>(pr + sp * split_stride + vo + j4);
          sum[c][j4] += f.x; sum[c][j4 + 1] += f.y; sum[c][j4 + 2] += f.z; sum[c][j4 + 3] += f.w;
        }
      }
    }
    const float xs = part_xs ? part_xs[row] : 1.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      const int vi = threadIdx.x + c * BT;
      const int64_t vo = (int64_t)(vi < nvec ? vi : 0) * VE;
      if (part_xs) {   // partials of an fp8 GEMM: (acc * x_scale[m]) * w_scale[n], its own epilogue
#pragma unroll
        for (int j = 0; j < VE; ++j) a[c].v[j] = from_f32<T>((sum[c][j] * xs) * part_ws[vo + j]);
      } else {
#pragma unroll
        for (int j = 0; j < VE; ++j) a[c].v[j] = from_f32<T>(sum[c][j] + 0.f);
      }
    }
  }
  // Phase 2: add, store the new residual, accumulate the sum of squares
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int vi = threadIdx.x + c * BT;
    if (vi < nvec) {
      if (ADD) {
        Vec<T, VE> s;
#pragma unroll
        for (int j = 0; j < VE; ++j) {
          float f = to_f32(a[c].v[j]) + to_f32(b[c].v[j]);
          vals[c][j] = f;
          s.v[j] = from_f32<T>(f);
        }
        stvec<T, VE>(rr + (int64_t)vi * VE, s);
      } else {
#pragma unroll
        for (int j = 0; j < VE; ++j) vals[c][j] = to_f32(a[c].v[j]);
      }
#pragma unroll
      for (int j = 0; j < VE; ++j) ss += vals[c][j] * vals[c][j];
    }
  }
  ss = block_sum(ss, red);
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
  Vec<T, VE> ov[CHUNKS];
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int vi = threadIdx.x + c * BT;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        ov[c].v[j] = from_f32<T>((vals[c][j] * inv) * to_f32(wv[c].v[j]));
        amax = fmaxf(amax, fabsf(to_f32(ov[c].v[j])));
      }
      stvec<T, VE>(orow + (int64_t)vi * VE, ov[c]);
    }
  }
  if constexpr (VE == 8) {
    if (q_out) {
      // per-token fp8 quantisation of the normalised row, for the fp8 GEMM that consumes it
      // (same arithmetic as per_token_quant_fp8 on the stored 16-bit row)
      amax = wave_max(amax);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
      __syncthreads();
      amax = red[0];
#pragma unroll
      for (int wv_ = 1; wv_ < BT / 64; ++wv_) amax = fmaxf(amax, red[wv_]);
      const float s = fmaxf(amax, 1e-10f) / 448.0f;
      const float qinv = 1.0f / s;
      if (threadIdx.x == 0) q_scale[row] = s;
#pragma unroll
      for (int c = 0; c < CHUNKS; ++c) {
        const int vi = threadIdx.x + c * BT;
        if (vi < nvec) *reinterpret_cast<uint2*>(q_out + row * hidden + (int64_t)vi * 8) = quant8_fp8<T>(ov[c], qinv);
      }
    }
  }
}

// Any hidden size / alignment; two passes over the row (second pass re-reads from L2).
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_generic_kernel(T* __restrict__ out,
                                                              const T* __restrict__ x,
                                                              const T* __restrict__ w,
                                                              int64_t hidden, int64_t x_stride,
                                                              int64_t out_stride, float eps) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const T* xr = x + row * x_stride;
  T* orow = out + row * out_stride;
  float ss = 0.f;
  for (int64_t i = threadIdx.x; i < hidden; i += 256) {
    const float f = to_f32(xr[i]);
    ss += f * f;
  }
  ss = block_sum(ss, red);
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
  for (int64_t i = threadIdx.x; i < hidden; i += 256)
    orow[i] = from_f32<T>((to_f32(xr[i]) * inv) * to_f32(w[i]));
}

// Generic ADD variant that keeps the unrounded fp32 sum: pass 1 computes the sum of squares
// from the inputs WITHOUT storing; pass 2 re-reads both inputs, stores residual and output.
template <typename T>
__global__ __launch_bounds__(256) void fused_add_rmsnorm_generic_kernel(T* __restrict__ x,
                                                                        T* __restrict__ residual,
                                                                        const T* __restrict__ w,
                                                                        int64_t hidden, float eps) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  T* xr = x + row * hidden;
  T* rr = residual + row * hidden;
  float ss = 0.f;
  for (int64_t i = threadIdx.x; i < hidden; i += 256) {
    float f = to_f32(xr[i]) + to_f32(rr[i]);
    ss += f * f;
  }
  ss = block_sum(ss, red);
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
  for (int64_t i = threadIdx.x; i < hidden; i += 256) {
    float f = to_f32(xr[i]) + to_f32(rr[i]);
    rr[i] = from_f32<T>(f);
    xr[i] = from_f32<T>((f * inv) * to_f32(w[i]));
  }
}

template <typename T, bool ADD>
static int launch_rmsnorm(T* out, T* x, T* residual, const T* w, int64_t rows, int64_t hidden,
                          int64_t x_stride, int64_t out_stride, float eps, hipStream_t st,
                          const float* part = nullptr, int nsplit = 0, const float* part_xs = nullptr,
                          const float* part_ws = nullptr, uint8_t* q_out = nullptr, float* q_scale = nullptr) {
  constexpr int VE = V16<T>::n;
  const bool vec_ok = hidden % VE == 0 && x_stride % VE == 0 && out_stride % VE == 0 &&
                      aligned16(out) && aligned16(x) && aligned16(w) &&
                      (!ADD || aligned16(residual));
  const int64_t nvec = hidden / VE;
  dim3 grid((unsigned)rows), block(256);
  if ((part || q_out) && !(vec_ok && nvec <= 256 * 8 && VE % 4 == 0)) return MI355_ERR_UNSUPPORTED;
  if (q_out && VE != 8) return MI355_ERR_UNSUPPORTED;
  if (vec_ok && nvec <= 256 * 8) {
    const int chunks = (int)((nvec + 255) / 256);
    WeightPrefetch pf;
    const int pf_wgs = take_prefetch(&pf);
    grid.x += pf_wgs;
#define LAUNCH(C, BT_)                                                                                  \
  hipLaunchKernelGGL((rmsnorm_vec_kernel<T, ADD, C, BT_>), grid, dim3(BT_), 0, st, out, x, residual, w, \
                     hidden, x_stride, out_stride, eps, pf, pf_wgs, part, nsplit, part_xs, part_ws, q_out, q_scale)
    // Rows of more than 256 vectors get 512 lanes (half the chunks per lane: all loads of a row, split-K
    // partials included, in fewer memory round trips).  The lane count is a function of the row length
    // alone, so a row's bits never depend on the batch it is normalised in.
    if (chunks <= 1) LAUNCH(1, 256);
    else if (chunks <= 2) LAUNCH(1, 512);
    else if (chunks <= 4) LAUNCH(2, 512);
    else LAUNCH(4, 512);
#undef LAUNCH
  } else if (ADD) {
    hipLaunchKernelGGL((fused_add_rmsnorm_generic_kernel<T>), grid, block, 0, st, x, residual, w,
                       hidden, eps);
  } else {
    hipLaunchKernelGGL((rmsnorm_generic_kernel<T>), grid, block, 0, st, out, (const T*)x, w,
                       hidden, x_stride, out_stride, eps);
  }
  return MI355_OK;
}

// ------------------------------------------------------------------------------------------
// SiLU-and-mul
// ------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void silu_and_mul_kernel(T* __restrict__ out,
                                                           const T* __restrict__ x, int64_t rows,
                                                           int64_t d) {
  constexpr int VE = VEC ? V16<T>::n : 1;
  const int64_t per_row = d / VE;
  const int64_t total = rows * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / per_row, c = (i - r * per_row) * VE;
    const T* g = x + r * 2 * d + c;
    Vec<T, VE> a = ldvec<T, VE>(g), b = ldvec<T, VE>(g + d), o;
#pragma unroll
    for (int j = 0; j < VE; ++j) o.v[j] = silu_mul_one<T>(a.v[j], b.v[j]);
    stvec<T, VE>(out + r * d + c, o);
  }
}

// Final-logit soft cap, in place on fp32 logits: x <- cap * tanh(x / cap), with tanh written as
// (e^{2y} - 1) / (e^{2y} + 1) the way the reference's kernel writes it
// (python/sglang/srt/layers/logits_processor.py:579-604).  The exponent is clamped at 40: beyond
// |y| = 10 the quotient is +-1 in fp32, and without the clamp e^{2y} overflows to inf / inf.
__device__ __forceinline__ float softcap_one(float x, float cap) {
  const float y2 = 2.f * (x / cap);
  const float y = y2 > 40.f ? 40.f : y2;   // not fminf: a NaN logit must stay NaN (fminf would turn it into +cap)
  const float e = expf(y);
  return (e - 1.f) / (e + 1.f) * cap;
}
template <bool VEC>
__global__ __launch_bounds__(256) void softcap_kernel(float* __restrict__ x, int64_t n, float cap) {
  constexpr int VE = VEC ? 4 : 1;
  const int64_t total = n / VE;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    Vec<float, VE> v = ldvec<float, VE>(x + i * VE);
#pragma unroll
    for (int j = 0; j < VE; ++j) v.v[j] = softcap_one(v.v[j], cap);
    stvec<float, VE>(x + i * VE, v);
  }
}

// SiLU-and-mul fused with the per-token fp8 quantisation of its output (the activation of an fp8
// down projection): one workgroup per row, the row stays in registers between the activation, the
// absmax and the conversion.  d % 8 == 0, d <= 16384 (1024 threads x 2 vectors of 8).  Same values as silu_and_mul followed by
// per_token_quant_fp8 on its 16-bit output.
template <typename T>
__global__ __launch_bounds__(1024) void silu_mul_quant_fp8_kernel(uint8_t* __restrict__ q_out,
                                                                  float* __restrict__ q_scale,
                                                                  const T* __restrict__ x, int64_t d,
                                                                  int64_t q_stride) {
  // 1024 threads per row: with one workgroup per token the launch is latency-bound, so the row is
  // spread over as many lanes as a workgroup has (two 16-byte pairs per thread at d = 14336).
  constexpr int MAXV = 2;
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const T* g = x + row * 2 * d;
  const int nvec = (int)(d >> 3);
  Vec<T, 8> o[MAXV];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = threadIdx.x + i * 1024;
    if (idx < nvec) {
      const Vec<T, 8> a = ldvec<T, 8>(g + (int64_t)idx * 8), b = ldvec<T, 8>(g + d + (int64_t)idx * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[i].v[j] = silu_mul_one<T>(a.v[j], b.v[j]);
        amax = fmaxf(amax, fabsf(to_f32(o[i].v[j])));
      }
    }
  }
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) amax = fmaxf(amax, red[w]);
  const float s = fmaxf(amax, 1e-10f) / 448.0f;
  const float inv = 1.0f / s;
  if (threadIdx.x == 0) q_scale[row] = s;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = threadIdx.x + i * 1024;
    if (idx < nvec) *reinterpret_cast<uint2*>(q_out + row * q_stride + (int64_t)idx * 8) = quant8_fp8<T>(o[i], inv);
  }
  // a row padded for the consumer (a reduction length rounded up to what its kernel streams): zeros behind d
  for (int idx = nvec + threadIdx.x; idx < (int)(q_stride >> 3); idx += 1024)
    *reinterpret_cast<uint2*>(q_out + row * q_stride + (int64_t)idx * 8) = uint2{0u, 0u};
}

// ------------------------------------------------------------------------------------------
// RoPE (+ fused KV-cache write).  One workgroup per token.  Work items: for every q and k head,
// rotary_dim/2 rotation pairs handled VE pairs at a time; then pass-through/k/v copies to the
// cache.  Rounding mirrors torch elementwise ops on dtype T: cos,sin -> T; each product and
// each sum rounds to T.
// ------------------------------------------------------------------------------------------
template <typename T, typename CT, typename LT, bool VEC = false>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ q, T* __restrict__ k,
                                                   const T* __restrict__ v,
                                                   const int64_t* __restrict__ positions,
                                                   const CT* __restrict__ cache, int hq, int hkv,
                                                   int head_dim, int rotary_dim, int64_t q_stride,
                                                   int64_t k_stride, int64_t v_stride, int is_neox,
                                                   T* __restrict__ k_buffer,
                                                   T* __restrict__ v_buffer,
                                                   const LT* __restrict__ loc, int64_t kv_slot, int64_t kv_head) {
  // kv_slot / kv_head: elements between consecutive slots / kv heads of the cache buffers.
  // Every product and sum below must round on its own (torch evaluates x1*cos, x2*sin and their
  // difference as separate elementwise ops): no fused multiply-add contraction in this kernel.
#pragma clang fp contract(off)
  // grid (tokens, chunks): chunk y < rot_chunks rotates 256 (head, pair) items, the remaining
  // chunks copy the pass-through part of k and all of v into the cache -- one memory round trip
  // per workgroup instead of a serial loop per token.
  const int64_t t = blockIdx.x;
  const int64_t pos = positions[t];
  const CT* cs = cache + pos * rotary_dim;
  const int half = rotary_dim / 2;
  T* qt = q + t * q_stride;
  T* kt = k + t * k_stride;
  const int nheads = hq + hkv;
  const int64_t slot = k_buffer ? (int64_t)loc[t] : 0;
  T* kdst = k_buffer ? k_buffer + slot * kv_slot : nullptr;
  const int rot_chunks = VEC ? (nheads * (half / V16<T>::n) + 255) / 256 : (nheads * half + 255) / 256;

  if ((int)blockIdx.y < rot_chunks) {
    if constexpr (VEC) {
      // neox layout, 16-byte accesses: one item = VE consecutive pairs (i, i + half) of one head
      constexpr int VE = V16<T>::n;
      const int per_head = half / VE;
      const int w = blockIdx.y * 256 + threadIdx.x;
      if (w >= nheads * per_head) return;
      const int h = w / per_head, i = (w - h * per_head) * VE;
      T* base = (h < hq) ? qt + (int64_t)h * head_dim : kt + (int64_t)(h - hq) * head_dim;
      const Vec<T, VE> a = ldvec<T, VE>(base + i), b = ldvec<T, VE>(base + i + half);
      Vec<T, VE> o1v, o2v;
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        const float c = round_to<T>((float)cs[i + j]), sn = round_to<T>((float)cs[half + i + j]);
        const float x1 = to_f32(a.v[j]), x2 = to_f32(b.v[j]);
        o1v.v[j] = from_f32<T>(round_to<T>(round_to<T>(x1 * c) - round_to<T>(x2 * sn)));
        o2v.v[j] = from_f32<T>(round_to<T>(round_to<T>(x2 * c) + round_to<T>(x1 * sn)));
      }
      stvec<T, VE>(base + i, o1v);
      stvec<T, VE>(base + i + half, o2v);
      if (kdst && h >= hq) {
        T* d = kdst + (int64_t)(h - hq) * kv_head;
        stvec<T, VE>(d + i, o1v);
        stvec<T, VE>(d + i + half, o2v);
      }
      return;
    }
    const int w = blockIdx.y * 256 + threadIdx.x;
    if (w >= nheads * half) return;
    const int h = w / half, i = w - h * half;
    T* base = (h < hq) ? qt + (int64_t)h * head_dim : kt + (int64_t)(h - hq) * head_dim;
    const int i1 = is_neox ? i : 2 * i, i2 = is_neox ? i + half : 2 * i + 1;
    const float c = round_to<T>((float)cs[i]), s = round_to<T>((float)cs[half + i]);
    const float x1 = to_f32(base[i1]), x2 = to_f32(base[i2]);
    const float o1 = round_to<T>(round_to<T>(x1 * c) - round_to<T>(x2 * s));
    const float o2 = round_to<T>(round_to<T>(x2 * c) + round_to<T>(x1 * s));
    base[i1] = from_f32<T>(o1);
    base[i2] = from_f32<T>(o2);
    if (kdst && h >= hq) {
      T* d = kdst + (int64_t)(h - hq) * kv_head;
      d[i1] = from_f32<T>(o1);
      d[i2] = from_f32<T>(o2);
    }
    return;
  }
  if (k_buffer) {
    // pass-through part of k (rotary_dim < head_dim) and all of v
    const int cw = (blockIdx.y - rot_chunks) * 256 + threadIdx.x;
    const int pass = head_dim - rotary_dim;
    if (cw < hkv * pass) {
      const int h = cw / pass, i = rotary_dim + (cw - h * pass);
      kdst[(int64_t)h * kv_head + i] = kt[(int64_t)h * head_dim + i];
    }
    if constexpr (VEC) {
      constexpr int VE = V16<T>::n;
      if (cw * VE < hkv * head_dim) {
        const int h = cw * VE / head_dim, d = cw * VE - h * head_dim;    // (VE divides head_dim on this path)
        stvec<T, VE>(v_buffer + slot * kv_slot + (int64_t)h * kv_head + d, ldvec<T, VE>(v + t * v_stride + cw * VE));
      }
    } else if (cw < hkv * head_dim) {
      const int h = cw / head_dim, d = cw - h * head_dim;
      v_buffer[slot * kv_slot + (int64_t)h * kv_head + d] = v[t * v_stride + cw];
    }
  }
}

// ------------------------------------------------------------------------------------------
// KV-cache scatter: one workgroup per token copies K row and V row to slot loc[t].
// ------------------------------------------------------------------------------------------
template <typename T, typename LT, bool VEC>
__global__ __launch_bounds__(256) void set_kv_kernel(T* __restrict__ kb, T* __restrict__ vb,
                                                     const T* __restrict__ k,
                                                     const T* __restrict__ v,
                                                     const LT* __restrict__ loc, int64_t k_row,
                                                     int64_t v_row, int64_t k_stride,
                                                     int64_t v_stride) {
  constexpr int VE = VEC ? V16<T>::n : 1;
  const int64_t t = blockIdx.x;
  const int64_t slot = (int64_t)loc[t];
  const T* ks = k + t * k_stride;
  const T* vs = v + t * v_stride;
  T* kd = kb + slot * k_row;
  T* vd = vb + slot * v_row;
  for (int64_t i = threadIdx.x * VE; i < k_row; i += 256 * VE)
    stvec<T, VE>(kd + i, ldvec<T, VE>(ks + i));
  for (int64_t i = threadIdx.x * VE; i < v_row; i += 256 * VE)
    stvec<T, VE>(vd + i, ldvec<T, VE>(vs + i));
}

// KV-cache scatter into an e4m3fn cache: rows are stored as clamp(x, +-448) rounded to nearest even
// (cache_k.to(float8_e4m3fn), mem_cache/memory_pool.py:436-446, saturating instead of overflowing).
// Rows are multiples of 8 elements.
template <typename T, typename LT>
__global__ __launch_bounds__(256) void set_kv_fp8_kernel(uint8_t* __restrict__ kb, uint8_t* __restrict__ vb,
                                                         const T* __restrict__ k, const T* __restrict__ v,
                                                         const LT* __restrict__ loc, int64_t k_row,
                                                         int64_t v_row, int64_t k_stride, int64_t v_stride) {
  const int64_t t = blockIdx.x;
  const int64_t slot = (int64_t)loc[t];
  const T* ks = k + t * k_stride;
  const T* vs = v + t * v_stride;
  for (int64_t i = threadIdx.x * 8; i < k_row; i += 256 * 8)
    *reinterpret_cast<uint2*>(kb + slot * k_row + i) = quant8_fp8<T>(ldvec<T, 8>(ks + i), 1.0f);
  for (int64_t i = threadIdx.x * 8; i < v_row; i += 256 * 8)
    *reinterpret_cast<uint2*>(vb + slot * v_row + i) = quant8_fp8<T>(ldvec<T, 8>(vs + i), 1.0f);
}

// KV-cache scatter by head: row t of k / v is `heads` heads of kd / vd features; head h goes to
// buffer + slot * slot_stride + h * head_stride (any layout kv_strides() describes), 16-bit or quantised to e4m3fn
// with set_kv_fp8_kernel's arithmetic.  kd, vd multiples of 8.
template <typename T, typename LT, bool FP8>
__global__ __launch_bounds__(256) void set_kv_heads_kernel(void* __restrict__ kb, void* __restrict__ vb,
                                                           const T* __restrict__ k, const T* __restrict__ v,
                                                           const LT* __restrict__ loc, int heads, int kd, int vd,
                                                           int64_t k_stride, int64_t v_stride, int64_t kb_slot,
                                                           int64_t kb_head, int64_t vb_slot, int64_t vb_head) {
  const int64_t t = blockIdx.x;
  const int64_t slot = (int64_t)loc[t];
  for (int i = threadIdx.x * 8; i < heads * kd; i += 256 * 8) {
    const int h = i / kd, d = i - h * kd;
    const Vec<T, 8> x = ldvec<T, 8>(k + t * k_stride + i);
    const int64_t off = slot * kb_slot + (int64_t)h * kb_head + d;
    if constexpr (FP8) *reinterpret_cast<uint2*>(static_cast<uint8_t*>(kb) + off) = quant8_fp8<T>(x, 1.0f);
    else stvec<T, 8>(static_cast<T*>(kb) + off, x);
  }
  for (int i = threadIdx.x * 8; i < heads * vd; i += 256 * 8) {
    const int h = i / vd, d = i - h * vd;
    const Vec<T, 8> x = ldvec<T, 8>(v + t * v_stride + i);
    const int64_t off = slot * vb_slot + (int64_t)h * vb_head + d;
    if constexpr (FP8) *reinterpret_cast<uint2*>(static_cast<uint8_t*>(vb) + off) = quant8_fp8<T>(x, 1.0f);
    else stvec<T, 8>(static_cast<T*>(vb) + off, x);
  }
}

// ------------------------------------------------------------------------------------------
// Embedding gather
// ------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void embedding_kernel(T* __restrict__ out,
                                                        const int64_t* __restrict__ ids,
                                                        const T* __restrict__ table,
                                                        int64_t hidden, int64_t v0, int64_t v1) {
  constexpr int VE = VEC ? V16<T>::n : 1;
  const int64_t t = blockIdx.x;
  const int64_t id = ids[t];
  const bool ok = id >= v0 && id < v1;
  const T* src = table + (ok ? (id - v0) : 0) * hidden;
  T* dst = out + t * hidden;
  for (int64_t i = threadIdx.x * VE; i < hidden; i += 256 * VE) {
    Vec<T, VE> x = ldvec<T, VE>(src + i);
    if (!ok) {
#pragma unroll
      for (int j = 0; j < VE; ++j) x.v[j] = from_f32<T>(0.f);
    }
    stvec<T, VE>(dst + i, x);
  }
}

template <typename T>
static void rope_launch(void* q, void* k, const void* v, const int64_t* positions,
                        const void* cache, int cache_dtype, int64_t tokens, int hq, int hkv,
                        int head_dim, int rotary_dim, int64_t q_stride, int64_t k_stride,
                        int64_t v_stride, int is_neox, void* k_buffer, void* v_buffer,
                        const void* loc, int loc_itype, int64_t kv_slot, int64_t kv_head, hipStream_t st) {
  constexpr int VE = V16<T>::n;
  // 16-byte path: neox pairs, whole head rotated, everything 16-byte aligned
  const bool vec = is_neox && rotary_dim == head_dim && (rotary_dim / 2) % VE == 0 && q_stride % VE == 0 &&
                   k_stride % VE == 0 && aligned16(q) && aligned16(k) &&
                   (!k_buffer || (v_stride % VE == 0 && aligned16(v) && aligned16(k_buffer) && aligned16(v_buffer)));
  const int rot_items = vec ? (hq + hkv) * (rotary_dim / 2 / VE) : (hq + hkv) * (rotary_dim / 2);
  const int copy_items = !k_buffer ? 0 : vec ? hkv * head_dim / VE : hkv * head_dim;
  const int rot_chunks = (rot_items + 255) / 256;
  const int copy_chunks = (copy_items + 255) / 256;
  const dim3 grid((unsigned)tokens, (unsigned)(rot_chunks + copy_chunks)), block(256);
  const bool f32cache = cache_dtype == MI355_F32 && DType<T>::code != MI355_F32;
  const bool i32 = loc_itype == MI355_I32;
#define ROPE_ARGS(CT, LT)                                                                    \
  (T*)q, (T*)k, (const T*)v, positions, (const CT*)cache, hq, hkv, head_dim, rotary_dim,      \
      q_stride, k_stride, v_stride, is_neox, (T*)k_buffer, (T*)v_buffer, (const LT*)loc, kv_slot, kv_head
#define ROPE_LAUNCH(CT, LT)                                                                              \
  do {                                                                                                   \
    if (vec) hipLaunchKernelGGL((rope_kernel<T, CT, LT, true>), grid, block, 0, st, ROPE_ARGS(CT, LT));  \
    else hipLaunchKernelGGL((rope_kernel<T, CT, LT, false>), grid, block, 0, st, ROPE_ARGS(CT, LT));     \
  } while (0)
  if (f32cache && i32) ROPE_LAUNCH(float, int32_t);
  else if (f32cache) ROPE_LAUNCH(float, int64_t);
  else if (i32) ROPE_LAUNCH(T, int32_t);
  else ROPE_LAUNCH(T, int64_t);
#undef ROPE_LAUNCH
#undef ROPE_ARGS
}

template <typename T>
static void set_kv_launch(void* kb, void* vb, const void* k, const void* v, const void* loc,
                          int loc_itype, int64_t tokens, int64_t k_row, int64_t v_row,
                          int64_t k_stride, int64_t v_stride, hipStream_t st) {
  constexpr int VE = V16<T>::n;
  const bool vec = k_row % VE == 0 && v_row % VE == 0 && k_stride % VE == 0 && v_stride % VE == 0 &&
                   aligned16(kb) && aligned16(vb) && aligned16(k) && aligned16(v);
  const dim3 grid((unsigned)tokens), block(256);
#define KV_ARGS(LT) (T*)kb, (T*)vb, (const T*)k, (const T*)v, (const LT*)loc, k_row, v_row, k_stride, v_stride
  if (loc_itype == MI355_I32) {
    if (vec) hipLaunchKernelGGL((set_kv_kernel<T, int32_t, true>), grid, block, 0, st, KV_ARGS(int32_t));
    else hipLaunchKernelGGL((set_kv_kernel<T, int32_t, false>), grid, block, 0, st, KV_ARGS(int32_t));
  } else {
    if (vec) hipLaunchKernelGGL((set_kv_kernel<T, int64_t, true>), grid, block, 0, st, KV_ARGS(int64_t));
    else hipLaunchKernelGGL((set_kv_kernel<T, int64_t, false>), grid, block, 0, st, KV_ARGS(int64_t));
  }
#undef KV_ARGS
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_abi_version(void) { return MI355_ABI_VERSION; }
const char* mi355_last_error(void) { return mi355::get_error(); }

int mi355_prefetch_next(const void* w_packed, int64_t m, int64_t n, int64_t k, int w_elem_bytes,
                        int64_t budget_bytes) {
  g_pf_pending = false;
  if (!w_packed) return MI355_OK;
  MI355_REQUIRE(w_elem_bytes == 2 || w_elem_bytes == 1, MI355_ERR_UNSUPPORTED,
                "prefetch_next: packed weights are 16-bit or fp8 (w_elem_bytes 2 or 1)");
  const int64_t k16 = w_elem_bytes == 1 ? k / 2 : k;
  g_pf_pending = plan_weight_prefetch(w_packed, m, n, k16, budget_bytes, &g_pf);
  return MI355_OK;
}

int mi355_fused_add_rmsnorm_partials(void* x, void* residual, const void* weight, const float* part,
                                     int nsplit, int64_t rows, int64_t hidden, float eps, int dtype,
                                     void* stream) {
  MI355_REQUIRE(x && residual && weight && part, MI355_ERR_INVALID, "fused_add_rmsnorm_partials: null pointer");
  MI355_REQUIRE(rows >= 0 && hidden > 0 && nsplit >= 1 && nsplit <= 64, MI355_ERR_INVALID,
                "fused_add_rmsnorm_partials: bad shape rows=%lld hidden=%lld nsplit=%d", (long long)rows,
                (long long)hidden, nsplit);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "fused_add_rmsnorm_partials: only 16-bit dtypes");
  MI355_REQUIRE(aligned16(part), MI355_ERR_INVALID, "fused_add_rmsnorm_partials: misaligned partials");
  if (rows == 0) return MI355_OK;
  int rc = MI355_OK;
  MI355_DISPATCH_DTYPE(dtype, "fused_add_rmsnorm_partials",
                       rc = launch_rmsnorm<T, true>((T*)x, (T*)x, (T*)residual, (const T*)weight, rows, hidden,
                                                    hidden, hidden, eps, (hipStream_t)stream, part, nsplit));
  MI355_REQUIRE(rc == MI355_OK, MI355_ERR_UNSUPPORTED,
                "fused_add_rmsnorm_partials: needs hidden %% 8 == 0, hidden <= 16384 and 16-B aligned pointers");
  MI355_CHECK_LAUNCH("fused_add_rmsnorm_partials");
  return MI355_OK;
}

int mi355_rmsnorm_ex(void* out, const void* x, void* residual, const void* weight, const float* part,
                     int nsplit, const float* part_x_scale, const float* part_w_scale, uint8_t* q_out,
                     float* q_scale, int64_t rows, int64_t hidden, float eps, int dtype, void* stream) {
  MI355_REQUIRE(out && weight && (x || part), MI355_ERR_INVALID, "rmsnorm_ex: null pointer");
  MI355_REQUIRE(rows >= 0 && hidden > 0, MI355_ERR_INVALID, "rmsnorm_ex: bad shape");
  MI355_REQUIRE(!part || (nsplit >= 1 && nsplit <= 64 && aligned16(part)), MI355_ERR_INVALID,
                "rmsnorm_ex: bad partials (nsplit=%d)", nsplit);
  MI355_REQUIRE((part_x_scale == nullptr) == (part_w_scale == nullptr) && (!part_x_scale || part), MI355_ERR_INVALID,
                "rmsnorm_ex: scales come in pairs and only with partials");
  MI355_REQUIRE((q_out == nullptr) == (q_scale == nullptr), MI355_ERR_INVALID, "rmsnorm_ex: q_out and q_scale come together");
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "rmsnorm_ex: only 16-bit dtypes");
  MI355_REQUIRE(!q_out || (reinterpret_cast<uintptr_t>(q_out) & 7) == 0, MI355_ERR_INVALID, "rmsnorm_ex: misaligned q_out");
  if (rows == 0) return MI355_OK;
  int rc = MI355_OK;
  const void* xin = x ? x : out;
  if (residual) {
    MI355_DISPATCH_DTYPE(dtype, "rmsnorm_ex",
                         rc = launch_rmsnorm<T, true>((T*)out, (T*)xin, (T*)residual, (const T*)weight, rows, hidden, hidden,
                                                      hidden, eps, (hipStream_t)stream, part, nsplit, part_x_scale,
                                                      part_w_scale, q_out, q_scale));
  } else {
    MI355_DISPATCH_DTYPE(dtype, "rmsnorm_ex",
                         rc = launch_rmsnorm<T, false>((T*)out, (T*)xin, nullptr, (const T*)weight, rows, hidden, hidden,
                                                       hidden, eps, (hipStream_t)stream, part, nsplit, part_x_scale,
                                                       part_w_scale, q_out, q_scale));
  }
  MI355_REQUIRE(rc == MI355_OK, MI355_ERR_UNSUPPORTED,
                "rmsnorm_ex: needs hidden %% 8 == 0, hidden <= 16384 and 16-B aligned pointers");
  MI355_CHECK_LAUNCH("rmsnorm_ex");
  return MI355_OK;
}

int mi355_silu_and_mul_quant_fp8(uint8_t* q_out, float* q_scale, const void* x, int64_t rows, int64_t d,
                                 int64_t q_stride, int dtype, void* stream) {
  MI355_REQUIRE(q_out && q_scale && x, MI355_ERR_INVALID, "silu_and_mul_quant_fp8: null pointer");
  MI355_REQUIRE(rows >= 0 && d > 0 && d % 8 == 0 && d <= 16384, MI355_ERR_UNSUPPORTED,
                "silu_and_mul_quant_fp8: needs d %% 8 == 0 and d <= 16384 (d=%lld)", (long long)d);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "silu_and_mul_quant_fp8: only 16-bit dtypes");
  MI355_REQUIRE(q_stride >= d && q_stride % 8 == 0, MI355_ERR_INVALID,
                "silu_and_mul_quant_fp8: q_stride must be a multiple of 8 and at least d (q_stride=%lld)", (long long)q_stride);
  MI355_REQUIRE(aligned16(x) && (reinterpret_cast<uintptr_t>(q_out) & 7) == 0, MI355_ERR_INVALID,
                "silu_and_mul_quant_fp8: misaligned pointer");
  if (rows == 0) return MI355_OK;
  if (dtype == MI355_BF16)
    hipLaunchKernelGGL((silu_mul_quant_fp8_kernel<bf16_t>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream,
                       q_out, q_scale, (const bf16_t*)x, d, q_stride);
  else
    hipLaunchKernelGGL((silu_mul_quant_fp8_kernel<f16_t>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream,
                       q_out, q_scale, (const f16_t*)x, d, q_stride);
  MI355_CHECK_LAUNCH("silu_and_mul_quant_fp8");
  return MI355_OK;
}

int mi355_rmsnorm(void* out, const void* x, const void* weight, int64_t rows, int64_t hidden,
                  int64_t x_row_stride, int64_t out_row_stride, float eps, int dtype,
                  void* stream) {
  MI355_REQUIRE(out && x && weight, MI355_ERR_INVALID, "rmsnorm: null pointer");
  MI355_REQUIRE(rows >= 0 && hidden > 0 && x_row_stride >= hidden && out_row_stride >= hidden,
                MI355_ERR_INVALID, "rmsnorm: bad shape rows=%lld hidden=%lld", (long long)rows,
                (long long)hidden);
  if (rows == 0) return MI355_OK;
  MI355_DISPATCH_DTYPE(dtype, "rmsnorm",
                       launch_rmsnorm<T, false>((T*)out, (T*)x, nullptr, (const T*)weight, rows,
                                                hidden, x_row_stride, out_row_stride, eps,
                                                (hipStream_t)stream));
  MI355_CHECK_LAUNCH("rmsnorm");
  return MI355_OK;
}

int mi355_fused_add_rmsnorm(void* x, void* residual, const void* weight, int64_t rows,
                            int64_t hidden, float eps, int dtype, void* stream) {
  MI355_REQUIRE(x && residual && weight, MI355_ERR_INVALID, "fused_add_rmsnorm: null pointer");
  MI355_REQUIRE(rows >= 0 && hidden > 0, MI355_ERR_INVALID, "fused_add_rmsnorm: bad shape");
  if (rows == 0) return MI355_OK;
  MI355_DISPATCH_DTYPE(dtype, "fused_add_rmsnorm",
                       launch_rmsnorm<T, true>((T*)x, (T*)x, (T*)residual, (const T*)weight, rows,
                                               hidden, hidden, hidden, eps, (hipStream_t)stream));
  MI355_CHECK_LAUNCH("fused_add_rmsnorm");
  return MI355_OK;
}

int mi355_silu_and_mul(void* out, const void* x, int64_t rows, int64_t d, int dtype,
                       void* stream) {
  MI355_REQUIRE(out && x, MI355_ERR_INVALID, "silu_and_mul: null pointer");
  MI355_REQUIRE(rows >= 0 && d > 0, MI355_ERR_INVALID, "silu_and_mul: bad shape");
  if (rows == 0) return MI355_OK;
  MI355_DISPATCH_DTYPE(dtype, "silu_and_mul", {
    constexpr int VE = V16<T>::n;
    const bool vec = d % VE == 0 && aligned16(out) && aligned16(x);
    const int64_t items = rows * (vec ? d / VE : d);
    const int grid = (int)std::min<int64_t>((items + 255) / 256, 256 * 16);
    if (vec)
      hipLaunchKernelGGL((silu_and_mul_kernel<T, true>), dim3(grid), dim3(256), 0,
                         (hipStream_t)stream, (T*)out, (const T*)x, rows, d);
    else
      hipLaunchKernelGGL((silu_and_mul_kernel<T, false>), dim3(grid), dim3(256), 0,
                         (hipStream_t)stream, (T*)out, (const T*)x, rows, d);
  });
  MI355_CHECK_LAUNCH("silu_and_mul");
  return MI355_OK;
}

int mi355_softcap_inplace(float* logits, int64_t n, float cap, void* stream) {
  MI355_REQUIRE(logits || n == 0, MI355_ERR_INVALID, "softcap_inplace: null pointer");
  MI355_REQUIRE(n >= 0 && cap > 0.f, MI355_ERR_INVALID, "softcap_inplace: bad size / cap");
  if (n == 0) return MI355_OK;
  const bool vec = n % 4 == 0 && aligned16(logits);
  const int64_t items = vec ? n / 4 : n;
  const int grid = (int)std::min<int64_t>((items + 255) / 256, 256 * 16);
  if (vec)
    hipLaunchKernelGGL((softcap_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, n, cap);
  else
    hipLaunchKernelGGL((softcap_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, n, cap);
  MI355_CHECK_LAUNCH("softcap_inplace");
  return MI355_OK;
}

int mi355_rope_inplace(void* q, void* k, const void* v, const int64_t* positions,
                       const void* cos_sin_cache, int cache_dtype, int64_t tokens, int hq,
                       int hkv, int head_dim, int rotary_dim, int64_t q_stride, int64_t k_stride,
                       int64_t v_stride, int is_neox, void* k_buffer, void* v_buffer,
                       const void* loc, int loc_itype, int64_t num_slots, int dtype,
                       void* stream) {
  const int kv_flags = dtype;
  dtype &= ~(MI355_KV_HEAD_MAJOR | MI355_KV_INTERLEAVED);
  MI355_REQUIRE(q && k && positions && cos_sin_cache, MI355_ERR_INVALID, "rope: null pointer");
  MI355_REQUIRE(tokens >= 0 && hq > 0 && hkv >= 0 && head_dim > 0 && rotary_dim > 0 &&
                    rotary_dim <= head_dim && rotary_dim % 2 == 0,
                MI355_ERR_INVALID, "rope: bad shape hq=%d hkv=%d head_dim=%d rotary_dim=%d", hq,
                hkv, head_dim, rotary_dim);
  MI355_REQUIRE(q_stride >= (int64_t)hq * head_dim && k_stride >= (int64_t)hkv * head_dim,
                MI355_ERR_INVALID, "rope: row strides smaller than rows");
  MI355_REQUIRE(cache_dtype == MI355_F32 || cache_dtype == dtype, MI355_ERR_UNSUPPORTED,
                "rope: cos_sin_cache must be fp32 or the tensor dtype");
  if (k_buffer) {
    MI355_REQUIRE(v && v_buffer && loc && num_slots > 0, MI355_ERR_INVALID,
                  "rope: fused KV write needs v, v_buffer, loc, num_slots");
    MI355_REQUIRE(v_stride >= (int64_t)hkv * head_dim, MI355_ERR_INVALID, "rope: bad v_stride");
  }
  if (tokens == 0) return MI355_OK;
  hipStream_t st = (hipStream_t)stream;
  int64_t kv_slot = 0, kv_head = 0;
  kv_strides(kv_flags, num_slots, hkv, head_dim, kv_slot, kv_head);
  MI355_DISPATCH_DTYPE(dtype, "rope",
                       rope_launch<T>(q, k, v, positions, cos_sin_cache, cache_dtype, tokens, hq,
                                      hkv, head_dim, rotary_dim, q_stride, k_stride, v_stride,
                                      is_neox, k_buffer, v_buffer, loc, loc_itype, kv_slot, kv_head, st));
  MI355_CHECK_LAUNCH("rope");
  return MI355_OK;
}

int mi355_set_kv_buffer(void* k_buffer, void* v_buffer, const void* k, const void* v,
                        const void* loc, int loc_itype, int64_t tokens, int64_t k_row_elems,
                        int64_t v_row_elems, int64_t k_stride, int64_t v_stride,
                        int64_t num_slots, int dtype, void* stream) {
  MI355_REQUIRE(k_buffer && v_buffer && k && v && loc, MI355_ERR_INVALID,
                "set_kv_buffer: null pointer");
  MI355_REQUIRE(!(dtype & MI355_KV_HEAD_MAJOR), MI355_ERR_UNSUPPORTED,
                "set_kv_buffer: rows are opaque here; a head-major cache is written by mi355_set_kv_buffer_heads");
  MI355_REQUIRE(tokens >= 0 && k_row_elems > 0 && v_row_elems > 0 && k_stride >= k_row_elems &&
                    v_stride >= v_row_elems && num_slots > 0,
                MI355_ERR_INVALID, "set_kv_buffer: bad shape");
  if (tokens == 0) return MI355_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype & MI355_KV_FP8) {   // e4m3fn cache, 16-bit k / v
    const int dt = dtype & 0xff;
    MI355_REQUIRE((dt == MI355_BF16 || dt == MI355_F16) && k_row_elems % 8 == 0 && v_row_elems % 8 == 0 &&
                      k_stride % 8 == 0 && v_stride % 8 == 0 && aligned16(k) && aligned16(v) &&
                      (reinterpret_cast<uintptr_t>(k_buffer) & 7) == 0 && (reinterpret_cast<uintptr_t>(v_buffer) & 7) == 0,
                  MI355_ERR_UNSUPPORTED, "set_kv_buffer: an fp8 cache needs 16-bit inputs, rows of 8-element multiples, aligned pointers");
    const dim3 grid((unsigned)tokens), block(256);
#define KV8_LAUNCH(T_, LT_)                                                                                 \
  hipLaunchKernelGGL((set_kv_fp8_kernel<T_, LT_>), grid, block, 0, st, (uint8_t*)k_buffer, (uint8_t*)v_buffer, \
                     (const T_*)k, (const T_*)v, (const LT_*)loc, k_row_elems, v_row_elems, k_stride, v_stride)
    if (dt == MI355_BF16) { if (loc_itype == MI355_I32) KV8_LAUNCH(bf16_t, int32_t); else KV8_LAUNCH(bf16_t, int64_t); }
    else { if (loc_itype == MI355_I32) KV8_LAUNCH(f16_t, int32_t); else KV8_LAUNCH(f16_t, int64_t); }
#undef KV8_LAUNCH
    MI355_CHECK_LAUNCH("set_kv_buffer");
    return MI355_OK;
  }
  MI355_DISPATCH_DTYPE(dtype, "set_kv_buffer",
                       set_kv_launch<T>(k_buffer, v_buffer, k, v, loc, loc_itype, tokens,
                                        k_row_elems, v_row_elems, k_stride, v_stride, st));
  MI355_CHECK_LAUNCH("set_kv_buffer");
  return MI355_OK;
}

int mi355_set_kv_buffer_heads(void* k_buffer, void* v_buffer, const void* k, const void* v, const void* loc,
                              int loc_itype, int64_t tokens, int heads, int k_head_dim, int v_head_dim, int64_t k_stride,
                              int64_t v_stride, int64_t num_slots, int dtype, void* stream) {
  MI355_REQUIRE(k_buffer && v_buffer && k && v && loc, MI355_ERR_INVALID, "set_kv_buffer_heads: null pointer");
  const bool kv8 = (dtype & MI355_KV_FP8) != 0;
  const int dt = dtype & 0xff;
  MI355_REQUIRE(tokens >= 0 && heads > 0 && k_head_dim > 0 && v_head_dim > 0 && num_slots > 0 &&
                    k_stride >= (int64_t)heads * k_head_dim && v_stride >= (int64_t)heads * v_head_dim,
                MI355_ERR_INVALID, "set_kv_buffer_heads: bad shape");
  MI355_REQUIRE(loc_itype == MI355_I32 || loc_itype == MI355_I64, MI355_ERR_INVALID, "set_kv_buffer_heads: loc must be int32 or int64");
  MI355_REQUIRE((dt == MI355_BF16 || dt == MI355_F16) && k_head_dim % 8 == 0 && v_head_dim % 8 == 0 && k_stride % 8 == 0 &&
                    v_stride % 8 == 0 && aligned16(k) && aligned16(v) &&
                    (reinterpret_cast<uintptr_t>(k_buffer) & (kv8 ? 7 : 15)) == 0 &&
                    (reinterpret_cast<uintptr_t>(v_buffer) & (kv8 ? 7 : 15)) == 0,
                MI355_ERR_UNSUPPORTED,
                "set_kv_buffer_heads: needs 16-bit k / v, head dims and row strides in multiples of 8, aligned pointers");
  MI355_REQUIRE(!(dtype & MI355_KV_INTERLEAVED) || ((dtype & MI355_KV_HEAD_MAJOR) && k_head_dim == v_head_dim), MI355_ERR_INVALID,
                "set_kv_buffer_heads: an interleaved cache is head-major with equal head dims");
  if (tokens == 0) return MI355_OK;
  int64_t ks, kh, vs, vh;
  kv_strides(dtype, num_slots, heads, k_head_dim, ks, kh);
  kv_strides(dtype, num_slots, heads, v_head_dim, vs, vh);
  const dim3 grid((unsigned)tokens), block(256);
  hipStream_t st = (hipStream_t)stream;
#define KVH_LAUNCH(T_, LT_, F8_)                                                                                   \
  hipLaunchKernelGGL((set_kv_heads_kernel<T_, LT_, F8_>), grid, block, 0, st, k_buffer, v_buffer, (const T_*)k,      \
                     (const T_*)v, (const LT_*)loc, heads, k_head_dim, v_head_dim, k_stride, v_stride, ks, kh, vs, vh)
#define KVH_LT(T_, F8_) do { if (loc_itype == MI355_I32) KVH_LAUNCH(T_, int32_t, F8_); else KVH_LAUNCH(T_, int64_t, F8_); } while (0)
  if (dt == MI355_BF16) { if (kv8) KVH_LT(bf16_t, true); else KVH_LT(bf16_t, false); }
  else { if (kv8) KVH_LT(f16_t, true); else KVH_LT(f16_t, false); }
#undef KVH_LT
#undef KVH_LAUNCH
  MI355_CHECK_LAUNCH("set_kv_buffer_heads");
  return MI355_OK;
}

int mi355_embedding(void* out, const int64_t* ids, const void* table, int64_t tokens,
                    int64_t hidden, int64_t vocab_start, int64_t vocab_end, int dtype,
                    void* stream) {
  MI355_REQUIRE(out && ids && table, MI355_ERR_INVALID, "embedding: null pointer");
  MI355_REQUIRE(tokens >= 0 && hidden > 0 && vocab_end > vocab_start, MI355_ERR_INVALID,
                "embedding: bad shape");
  if (tokens == 0) return MI355_OK;
  hipStream_t st = (hipStream_t)stream;
  MI355_DISPATCH_DTYPE(dtype, "embedding", {
    constexpr int VE = V16<T>::n;
    const bool vec = hidden % VE == 0 && aligned16(out) && aligned16(table);
    if (vec)
      hipLaunchKernelGGL((embedding_kernel<T, true>), dim3((unsigned)tokens), dim3(256), 0, st,
                         (T*)out, ids, (const T*)table, hidden, vocab_start, vocab_end);
    else
      hipLaunchKernelGGL((embedding_kernel<T, false>), dim3((unsigned)tokens), dim3(256), 0, st,
                         (T*)out, ids, (const T*)table, hidden, vocab_start, vocab_end);
  });
  MI355_CHECK_LAUNCH("embedding");
  return MI355_OK;
}

}  // extern "C"
