// fp8 W8A8 linear for gfx950: OCP e4m3fn (max 448) -- NOT the gfx942 fnuz encoding.
//  * per-token dynamic activation quantisation (one workgroup per row);
//  * y = (xq @ wq^T) * x_scale[m] * w_scale[n] on the fp8 MFMA (v_mfma_f32_16x16x32_fp8_fp8,
//    fp32 accumulate).  M <= 64 uses the same weight-streaming structure as the bf16 skinny
//    GEMM (weights HBM -> VGPR fragments, activation tile through LDS, split-K slabs); any
//    other shape uses an exact tiled fallback that decodes fp8 to fp32.
#include "common.cuh"
#include <algorithm>

namespace mi355 {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr float kFp8Max = 448.0f;

__device__ __forceinline__ float fp8_to_f32(uint8_t b) {
  return __builtin_amdgcn_cvt_f32_fp8((int)b, 0);
}

// ---- per-token quantisation ----------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void per_token_quant_fp8_kernel(uint8_t* __restrict__ xq,
                                                                  float* __restrict__ scales,
                                                                  const T* __restrict__ x, int64_t k,
                                                                  int64_t x_stride, int64_t q_stride) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const T* xr = x + row * x_stride;
  uint8_t* qr = xq + row * q_stride;
  for (int64_t i = k + threadIdx.x; i < q_stride; i += 256) qr[i] = 0;   // padding behind the row
  float amax = 0.f;
  for (int64_t i = threadIdx.x; i < k; i += 256) amax = fmaxf(amax, fabsf(to_f32(xr[i])));
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float s = fmaxf(amax, 1e-10f) / kFp8Max;
  const float inv = 1.0f / s;
  if (threadIdx.x == 0) scales[row] = s;
  // two elements per thread per step so that v_cvt_pk_fp8_f32 packs a pair
  for (int64_t i = 2 * (int64_t)threadIdx.x; i < k; i += 512) {
    const float a = fminf(fmaxf(to_f32(xr[i]) * inv, -kFp8Max), kFp8Max);
    const float b = i + 1 < k ? fminf(fmaxf(to_f32(xr[i + 1]) * inv, -kFp8Max), kFp8Max) : 0.f;
    const int packed = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    qr[i] = (uint8_t)(packed & 0xff);
    if (i + 1 < k) qr[i + 1] = (uint8_t)((packed >> 8) & 0xff);
  }
}

// Vectorised form for 16-bit inputs with k % 8 == 0, k <= 16384: the row is read ONCE with
// 16-byte loads and stays in registers between the absmax pass and the conversion (one memory
// round trip, 8-byte stores).  Same arithmetic, element by element, as the kernel above.
template <typename T>
__global__ __launch_bounds__(256) void per_token_quant_fp8_vec_kernel(uint8_t* __restrict__ xq,
                                                                      float* __restrict__ scales,
                                                                      const T* __restrict__ x, int64_t k,
                                                                      int64_t x_stride, int64_t q_stride) {
  constexpr int MAXV = 8;
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const T* xr = x + row * x_stride;
  uint8_t* qr = xq + row * q_stride;
  const int nvec = (int)(k >> 3);
  for (int idx = nvec + threadIdx.x; idx < (int)(q_stride >> 3); idx += 256)   // padding behind the row
    *reinterpret_cast<uint2*>(qr + (int64_t)idx * 8) = uint2{0u, 0u};
  Vec<T, 8> v[MAXV];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = threadIdx.x + i * 256;
    if (idx < nvec) {
      v[i] = ldvec<T, 8>(xr + (int64_t)idx * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(to_f32(v[i].v[j])));
    }
  }
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float s = fmaxf(amax, 1e-10f) / kFp8Max;
  const float inv = 1.0f / s;
  if (threadIdx.x == 0) scales[row] = s;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = threadIdx.x + i * 256;
    if (idx < nvec) {
      unsigned out[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = fminf(fmaxf(to_f32(v[i].v[4 * h + j]) * inv, -kFp8Max), kFp8Max);
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], pk, true);
        out[h] = (unsigned)pk;
      }
      *reinterpret_cast<uint2*>(qr + (int64_t)idx * 8) = uint2{out[0], out[1]};
    }
  }
}

// ---- skinny fp8 GEMM -------------------------------------------------------------------------
constexpr int F8_BK = 512;             // fp8 elements (bytes) of K per chunk
constexpr int F8_LDS_ROW = F8_BK + 16; // padded LDS row in bytes

template <typename OUT> __device__ __forceinline__ OUT f8_finish(float v) { return (OUT)v; }

// grid (ceil(N/64), splits), block 256.  Each lane's 16-byte weight load covers two MFMA
// k-steps: within a 64-wide K block, lane group q owns bytes [16q, 16q+16); step s uses its
// bytes [8s, 8s+8).  The activation fragment uses the same map, so products pair up.
template <int MT, typename OUT>
__global__ __launch_bounds__(256) void fp8_skinny_kernel(const uint8_t* __restrict__ xq,
                                                         const uint8_t* __restrict__ wq,
                                                         const float* __restrict__ x_scale,
                                                         const float* __restrict__ w_scale,
                                                         const OUT* __restrict__ bias,
                                                         OUT* __restrict__ y,
                                                         float* __restrict__ slab, int M, int N,
                                                         int K, int k_per_split) {
  constexpr int R = 16 * MT;
  constexpr int XP = R * (F8_BK / 16) / 256;  // 16-B pieces per thread (R/8)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint8_t* xs = reinterpret_cast<uint8_t*>(smem_raw);  // [2][R][F8_LDS_ROW]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int n_wave = blockIdx.x * 64 + wave * 16;
  const int ks = blockIdx.y * k_per_split;
  const int ke = min(K, ks + k_per_split);
  const int nchunks = (ke - ks + F8_BK - 1) / F8_BK;
  int n = n_wave + r16;
  n = n < N ? n : N - 1;
  const uint8_t* wrow = wq + (int64_t)n * K + q * 16;

  f32x4_t acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint4 wcur[8], wnext[8], xreg[XP];

  auto load_w = [&](uint4 (&dst)[8], int kc) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int k = kc + b * 64 + q * 16;
      dst[b] = k < ke ? *reinterpret_cast<const uint4*>(wrow + kc + b * 64) : uint4{0, 0, 0, 0};
    }
  };
  auto load_x = [&](int kc) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int p = threadIdx.x + i * 256;
      const int row = p / (F8_BK / 16), c16 = p % (F8_BK / 16);
      const int k = kc + c16 * 16;
      xreg[i] = (row < M && k < ke) ? *reinterpret_cast<const uint4*>(xq + (int64_t)row * K + k)
                                    : uint4{0, 0, 0, 0};
    }
  };
  auto store_x = [&](int buf) {
    uint8_t* dst = xs + buf * (R * F8_LDS_ROW);
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int p = threadIdx.x + i * 256;
      const int row = p / (F8_BK / 16), c16 = p % (F8_BK / 16);
      *reinterpret_cast<uint4*>(dst + row * F8_LDS_ROW + c16 * 16) = xreg[i];
    }
  };

  load_w(wcur, ks);
  load_x(ks);
  store_x(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const bool has_next = c + 1 < nchunks;
    if (has_next) {
      load_w(wnext, ks + (c + 1) * F8_BK);
      load_x(ks + (c + 1) * F8_BK);
    }
    const uint8_t* xt = xs + (c & 1) * (R * F8_LDS_ROW) + r16 * F8_LDS_ROW + q * 16;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      uint4 xf[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        xf[mt] = *reinterpret_cast<const uint4*>(xt + mt * 16 * F8_LDS_ROW + b * 64);
      const long w0 = (long)(((unsigned long)wcur[b].y << 32) | wcur[b].x);
      const long w1 = (long)(((unsigned long)wcur[b].w << 32) | wcur[b].z);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const long x0 = (long)(((unsigned long)xf[mt].y << 32) | xf[mt].x);
        const long x1 = (long)(((unsigned long)xf[mt].w << 32) | xf[mt].z);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w0, x0, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w1, x1, acc[mt], 0, 0, 0);
      }
    }
    if (has_next) {
      store_x((c + 1) & 1);
#pragma unroll
      for (int b = 0; b < 8; ++b) wcur[b] = wnext[b];
    }
    __syncthreads();
  }
  const int n0 = n_wave + q * 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + r16;
    if (m >= M) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nn = n0 + r;
      if (nn >= N) continue;
      if (slab) {
        slab[((int64_t)blockIdx.y * M + m) * N + nn] = acc[mt][r];
      } else {
        float v = (acc[mt][r] * x_scale[m]) * w_scale[nn];
        if (bias) v += (float)bias[nn];
        y[(int64_t)m * N + nn] = f8_finish<OUT>(v);
      }
    }
  }
}

template <typename OUT>
__global__ __launch_bounds__(256) void fp8_splitk_reduce_kernel(const float* __restrict__ slab,
                                                                const float* __restrict__ x_scale,
                                                                const float* __restrict__ w_scale,
                                                                const OUT* __restrict__ bias,
                                                                OUT* __restrict__ y, int M, int N,
                                                                int splits) {
  const int64_t mn = (int64_t)M * N;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= mn) return;
  float a = 0.f;
  for (int s = 0; s < splits; ++s) a += slab[(int64_t)s * mn + i];
  const int m = (int)(i / N), n = (int)(i % N);
  float v = (a * x_scale[m]) * w_scale[n];
  if (bias) v += (float)bias[n];
  y[i] = f8_finish<OUT>(v);
}

// exact fallback: one thread per output, fp32 fma chain over decoded fp8
template <typename OUT>
__global__ __launch_bounds__(256) void fp8_mm_generic_kernel(const uint8_t* __restrict__ xq,
                                                             const uint8_t* __restrict__ wq,
                                                             const float* __restrict__ x_scale,
                                                             const float* __restrict__ w_scale,
                                                             const OUT* __restrict__ bias,
                                                             OUT* __restrict__ y, int64_t M,
                                                             int64_t N, int64_t K) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i % N;
  const uint8_t* xr = xq + m * K;
  const uint8_t* wr = wq + n * K;
  float a = 0.f;
  for (int64_t k = 0; k < K; ++k) a = fmaf(fp8_to_f32(xr[k]), fp8_to_f32(wr[k]), a);
  float v = (a * x_scale[m]) * w_scale[n];
  if (bias) v += (float)bias[n];
  y[i] = f8_finish<OUT>(v);
}

struct F8Plan { int splits, k_per_split; };
static F8Plan plan_fp8(int64_t n, int64_t k) {
  const int64_t wg_n = (n + 63) / 64;
  int64_t splits = (512 + wg_n - 1) / wg_n;
  splits = std::max<int64_t>(1, std::min<int64_t>(splits, std::max<int64_t>(1, k / 1024)));
  int64_t kps = (k + splits - 1) / splits;
  kps = (kps + F8_BK - 1) / F8_BK * F8_BK;
  F8Plan p;
  p.k_per_split = (int)kps;
  p.splits = (int)((k + kps - 1) / kps);
  return p;
}

template <typename OUT>
static void launch_fp8_mm(void* y, const uint8_t* xq, const uint8_t* wq, const float* xs,
                          const float* ws, const void* bias, int64_t m, int64_t n, int64_t k,
                          void* workspace, bool skinny, hipStream_t st) {
  if (!skinny) {
    hipLaunchKernelGGL((fp8_mm_generic_kernel<OUT>), dim3((unsigned)((m * n + 255) / 256)),
                       dim3(256), 0, st, xq, wq, xs, ws, (const OUT*)bias, (OUT*)y, m, n, k);
    return;
  }
  const F8Plan p = plan_fp8(n, k);
  float* slab = p.splits > 1 ? (float*)workspace : nullptr;
  const dim3 grid((unsigned)((n + 63) / 64), (unsigned)p.splits);
  const int mt = (int)((m + 15) / 16);
#define F8(MT_)                                                                              \
  hipLaunchKernelGGL((fp8_skinny_kernel<MT_, OUT>), grid, dim3(256),                          \
                     (size_t)2 * 16 * MT_ * F8_LDS_ROW, st, xq, wq, xs, ws, (const OUT*)bias, \
                     (OUT*)y, slab, (int)m, (int)n, (int)k, p.k_per_split)
  if (mt <= 1) F8(1); else if (mt <= 2) F8(2); else F8(4);
#undef F8
  if (p.splits > 1)
    hipLaunchKernelGGL((fp8_splitk_reduce_kernel<OUT>), dim3((unsigned)((m * n + 255) / 256)),
                       dim3(256), 0, st, slab, xs, ws, (const OUT*)bias, (OUT*)y, (int)m, (int)n,
                       p.splits);
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_per_token_quant_fp8(uint8_t* xq, float* scales, const void* x, int64_t rows, int64_t k,
                              int64_t x_stride, int64_t q_stride, int dtype, void* stream) {
  MI355_REQUIRE(xq && scales && x, MI355_ERR_INVALID, "per_token_quant_fp8: null pointer");
  MI355_REQUIRE(rows >= 0 && k > 0 && x_stride >= k && q_stride >= k, MI355_ERR_INVALID,
                "per_token_quant_fp8: bad shape");
  if (rows == 0) return MI355_OK;
  const bool vec = dtype != MI355_F32 && k % 8 == 0 && k <= 16384 && x_stride % 8 == 0 && q_stride % 8 == 0 &&
                   aligned16(x) && (reinterpret_cast<uintptr_t>(xq) & 7) == 0;
  MI355_DISPATCH_DTYPE(dtype, "per_token_quant_fp8", {
    if (vec)
      hipLaunchKernelGGL((per_token_quant_fp8_vec_kernel<T>), dim3((unsigned)rows), dim3(256), 0,
                         (hipStream_t)stream, xq, scales, (const T*)x, k, x_stride, q_stride);
    else
      hipLaunchKernelGGL((per_token_quant_fp8_kernel<T>), dim3((unsigned)rows), dim3(256), 0,
                         (hipStream_t)stream, xq, scales, (const T*)x, k, x_stride, q_stride);
  });
  MI355_CHECK_LAUNCH("per_token_quant_fp8");
  return MI355_OK;
}

int64_t mi355_fp8_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m < 1 || m > 64) return 0;
  const F8Plan p = plan_fp8(n, k);
  return p.splits > 1 ? (int64_t)p.splits * m * n * 4 : 0;
}

int mi355_fp8_scaled_mm(void* y, const uint8_t* xq, const uint8_t* wq, const float* x_scale,
                        const float* w_scale, const void* bias, int64_t m, int64_t n, int64_t k,
                        void* workspace, int64_t workspace_bytes, int out_dtype, void* stream) {
  MI355_REQUIRE(y && xq && wq && x_scale && w_scale, MI355_ERR_INVALID,
                "fp8_scaled_mm: null pointer");
  MI355_REQUIRE(m >= 0 && n > 0 && k > 0 && n < (1LL << 31) && k < (1LL << 31), MI355_ERR_INVALID,
                "fp8_scaled_mm: bad shape");
  if (m == 0) return MI355_OK;
  const bool skinny = m <= 64 && k % 16 == 0 && aligned16(xq) && aligned16(wq);
  if (skinny) {
    const int64_t need = mi355_fp8_workspace_bytes(m, n, k);
    if (need > 0 && (!workspace || workspace_bytes < need)) {
      set_error("fp8_scaled_mm: workspace of %lld bytes required, got %lld", (long long)need,
                (long long)workspace_bytes);
      return MI355_ERR_INVALID;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  switch (out_dtype) {
    case MI355_F32: launch_fp8_mm<float>(y, xq, wq, x_scale, w_scale, bias, m, n, k, workspace, skinny, st); break;
    case MI355_BF16: launch_fp8_mm<bf16_t>(y, xq, wq, x_scale, w_scale, bias, m, n, k, workspace, skinny, st); break;
    case MI355_F16: launch_fp8_mm<f16_t>(y, xq, wq, x_scale, w_scale, bias, m, n, k, workspace, skinny, st); break;
    default:
      set_error("fp8_scaled_mm: unsupported out_dtype %d", out_dtype);
      return MI355_ERR_UNSUPPORTED;
  }
  MI355_CHECK_LAUNCH("fp8_scaled_mm");
  return MI355_OK;
}

}  // extern "C"
