// Dense projections y = x @ W^T (W in nn.Linear layout [N, K]).
//
//  * skinny path (M <= 64, bf16/f16): decode-time GEMMs are weight-streaming and HBM-bound.
//    Each wave streams its 16*NT weight rows straight from HBM into VGPRs as MFMA A-operand
//    fragments (16-B loads, a whole 256-deep K chunk prefetched one chunk ahead), the small
//    activation tile is staged once per workgroup through double-buffered LDS and re-used by
//    the 4 waves, and v_mfma_f32_16x16x32 accumulates in fp32.  The K axis is split across
//    workgroups (grid.y) so that small-N projections still fill 256 CUs; split partials go to
//    an fp32 slab and are summed in a fixed order by a tiny second kernel (deterministic).
//  * tiled path (any M, any dtype): exact-fp32-accumulate LDS-tiled VALU kernel.  It is the
//    fp32 parity path and the fallback for shapes the skinny kernel does not take.
#include "common.cuh"
#include <algorithm>

namespace mi355 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// Epilogue store of one value with optional bias; OUT is T or float (float = T-rounded).
template <typename T, typename OUT> __device__ __forceinline__ OUT finish(float acc, float bias) {
  return (OUT)round_to<T>(acc + bias);
}

constexpr int SK_BK = 256;            // K elements per chunk
constexpr int SK_LDS_ROW = SK_BK + 8; // padded LDS row (elements): 528 B rows spread banks

// grid: (ceil(N / (64*NT)), splits).  block: 256.
template <typename T, int MT, int NT, typename OUT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ bias,
    OUT* __restrict__ y, float* __restrict__ slab, int M, int N, int K, int64_t x_stride,
    int k_per_split) {
  using frag = typename Mfma<T>::frag;
  constexpr int R = 16 * MT;                  // activation rows held in LDS
  constexpr int XP = R * (SK_BK / 8) / 256;   // 16-B pieces of the x tile per thread (R/8)
  static_assert(XP >= 1, "tile too small");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* xs = reinterpret_cast<T*>(smem_raw);     // [2][R][SK_LDS_ROW]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int n_wave = blockIdx.x * (64 * NT) + wave * (16 * NT);
  const int ks = blockIdx.y * k_per_split;
  const int ke = min(K, ks + k_per_split);
  const int nchunks = (ke - ks + SK_BK - 1) / SK_BK;

  // weight row pointers (clamped: rows >= N are loaded from row N-1 and never stored)
  const T* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int n = n_wave + nt * 16 + r16;
    n = n < N ? n : N - 1;
    wrow[nt] = w + (int64_t)n * K + q * 8;
  }

  f32x4 acc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  frag wcur[NT][8], wnext[NT][8];
  Vec<T, 8> xreg[XP];

  auto load_w = [&](frag (&dst)[NT][8], int kc) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int k = kc + s * 32 + q * 8;
        if (k < ke) {
          dst[nt][s] = *reinterpret_cast<const frag*>(wrow[nt] + kc + s * 32);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) dst[nt][s][j] = (T)0.f;
        }
      }
  };
  auto load_x = [&](int kc) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int p = threadIdx.x + i * 256;
      const int row = p / (SK_BK / 8), c16 = p % (SK_BK / 8);
      const int k = kc + c16 * 8;
      if (row < M && k < ke) {
        xreg[i] = ldvec<T, 8>(x + (int64_t)row * x_stride + k);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xreg[i].v[j] = (T)0.f;
      }
    }
  };
  auto store_x = [&](int buf) {
    T* dst = xs + buf * (R * SK_LDS_ROW);
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int p = threadIdx.x + i * 256;
      const int row = p / (SK_BK / 8), c16 = p % (SK_BK / 8);
      stvec<T, 8>(dst + row * SK_LDS_ROW + c16 * 8, xreg[i]);
    }
  };

  load_w(wcur, ks);
  load_x(ks);
  store_x(0);
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int kc_next = ks + (c + 1) * SK_BK;
    const bool has_next = c + 1 < nchunks;
    if (has_next) {
      load_w(wnext, kc_next);
      load_x(kc_next);
    }
    const T* xt = xs + (c & 1) * (R * SK_LDS_ROW) + r16 * SK_LDS_ROW + q * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      frag xf[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        xf[mt] = *reinterpret_cast<const frag*>(xt + mt * 16 * SK_LDS_ROW + s * 32);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = Mfma<T>::run(wcur[nt][s], xf[mt], acc[nt][mt]);
    }
    if (has_next) {
      store_x((c + 1) & 1);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 8; ++s) wcur[nt][s] = wnext[nt][s];
    }
    __syncthreads();
  }

  // D[i = n (4q + r)][j = m (r16)]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n0 = n_wave + nt * 16 + q * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + r16;
      if (m >= M) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + r;
        if (n >= N) continue;
        if (slab) {
          slab[((int64_t)blockIdx.y * M + m) * N + n] = acc[nt][mt][r];
        } else {
          y[(int64_t)m * N + n] = finish<T, OUT>(acc[nt][mt][r], bias ? to_f32(bias[n]) : 0.f);
        }
      }
    }
  }
}

template <typename T, typename OUT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab,
                                                            const T* __restrict__ bias,
                                                            OUT* __restrict__ y, int64_t mn, int N,
                                                            int splits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= mn) return;
  float a = 0.f;
  for (int s = 0; s < splits; ++s) a += slab[(int64_t)s * mn + i];
  y[i] = finish<T, OUT>(a, bias ? to_f32(bias[i % N]) : 0.f);
}

// ------------------------------------------------------------------------------------------
// Tiled exact-fp32 path: 64x64 output tile per workgroup, BK = 16, 4x4 outputs per thread.
// ------------------------------------------------------------------------------------------
template <typename T, typename OUT>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(const T* __restrict__ x,
                                                         const T* __restrict__ w,
                                                         const T* __restrict__ bias,
                                                         OUT* __restrict__ y, int64_t M, int64_t N,
                                                         int64_t K, int64_t x_stride) {
  constexpr int BM = 64, BN = 64, BK = 16;
  __shared__ float xs[BK][BM + 4];
  __shared__ float ws[BK][BN + 4];
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // tx -> n, ty -> m
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int64_t k0 = 0; k0 < K; k0 += BK) {
    // each thread loads 4 x elements and 4 w elements: element e = tid + i*256 -> (row, k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = threadIdx.x + i * 256;
      const int row = e >> 4, kk = e & 15;
      const int64_t k = k0 + kk;
      xs[kk][row] = (m0 + row < M && k < K) ? to_f32(x[(m0 + row) * x_stride + k]) : 0.f;
      ws[kk][row] = (n0 + row < N && k < K) ? to_f32(w[(n0 + row) * K + k]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = xs[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = ws[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + tx * 4 + j;
      if (n < N) y[m * N + n] = finish<T, OUT>(acc[i][j], bias ? to_f32(bias[n]) : 0.f);
    }
  }
}

struct SkinnyPlan {
  int nt, splits, k_per_split;
};

static SkinnyPlan plan_skinny(int64_t n, int64_t k) {
  SkinnyPlan p;
  p.nt = n >= 64 * 1024 ? 2 : 1;  // very wide outputs (lm_head): 128 columns per workgroup
  const int64_t wg_n = (n + 64 * p.nt - 1) / (64 * p.nt);
  int64_t splits = (512 + wg_n - 1) / wg_n;
  const int64_t max_splits = std::max<int64_t>(1, k / 512);
  splits = std::min(splits, max_splits);
  splits = std::max<int64_t>(splits, 1);
  int64_t kps = (k + splits - 1) / splits;
  kps = (kps + SK_BK - 1) / SK_BK * SK_BK;
  p.k_per_split = (int)kps;
  p.splits = (int)((k + kps - 1) / kps);
  return p;
}

static bool skinny_ok(int64_t m, int64_t n, int64_t k, int64_t x_stride, const void* x,
                      const void* w, int dtype) {
  return (dtype == MI355_BF16 || dtype == MI355_F16) && m >= 1 && m <= 64 && k % 8 == 0 &&
         x_stride % 8 == 0 && aligned16(x) && aligned16(w) && n >= 1 && k >= 8;
}

template <typename T, int MT, int NT, typename OUT>
static void launch_skinny_cfg(const T* x, const T* w, const T* bias, OUT* y, float* slab, int M,
                              int N, int K, int64_t x_stride, const SkinnyPlan& p,
                              hipStream_t st) {
  const dim3 grid((unsigned)((N + 64 * NT - 1) / (64 * NT)), (unsigned)p.splits);
  const size_t lds = 2 * (16 * MT) * SK_LDS_ROW * sizeof(T);
  hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, NT, OUT>), grid, dim3(256), lds, st, x, w, bias, y,
                     slab, M, N, K, x_stride, p.k_per_split);
}

template <typename T, typename OUT>
static void launch_skinny(const T* x, const T* w, const T* bias, OUT* y, float* ws, int M, int N,
                          int K, int64_t x_stride, hipStream_t st) {
  const SkinnyPlan p = plan_skinny(N, K);
  float* slab = p.splits > 1 ? ws : nullptr;
  const int mt = (M + 15) / 16;
#define SK(MT_, NT_) launch_skinny_cfg<T, MT_, NT_, OUT>(x, w, bias, y, slab, M, N, K, x_stride, p, st)
  if (p.nt == 2) {
    if (mt <= 1) SK(1, 2); else if (mt <= 2) SK(2, 2); else SK(4, 2);
  } else {
    if (mt <= 1) SK(1, 1); else if (mt <= 2) SK(2, 1); else SK(4, 1);
  }
#undef SK
  if (p.splits > 1) {
    const int64_t mn = (int64_t)M * N;
    hipLaunchKernelGGL((splitk_reduce_kernel<T, OUT>), dim3((unsigned)((mn + 255) / 256)),
                       dim3(256), 0, st, slab, bias, y, mn, N, p.splits);
  }
}

template <typename T, typename OUT>
static void launch_linear_t(void* y, const void* x, const void* w, const void* bias, int64_t m,
                            int64_t n, int64_t k, int64_t x_stride, void* ws, bool skinny,
                            hipStream_t st) {
  if constexpr (!std::is_same<T, float>::value) {
    if (skinny) {
      launch_skinny<T, OUT>((const T*)x, (const T*)w, (const T*)bias, (OUT*)y, (float*)ws, (int)m,
                            (int)n, (int)k, x_stride, st);
      return;
    }
  }
  const dim3 grid((unsigned)((n + 63) / 64), (unsigned)((m + 63) / 64));
  hipLaunchKernelGGL((gemm_tiled_kernel<T, OUT>), grid, dim3(256), 0, st, (const T*)x,
                     (const T*)w, (const T*)bias, (OUT*)y, m, n, k, x_stride);
}

template <typename T>
static void launch_linear(void* y, const void* x, const void* w, const void* bias, int64_t m,
                          int64_t n, int64_t k, int64_t x_stride, void* ws, bool skinny,
                          bool out_f32, hipStream_t st) {
  if (out_f32) launch_linear_t<T, float>(y, x, w, bias, m, n, k, x_stride, ws, skinny, st);
  else launch_linear_t<T, T>(y, x, w, bias, m, n, k, x_stride, ws, skinny, st);
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int64_t mi355_linear_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m < 1 || m > 64 || n < 1 || k < 1) return 0;
  const SkinnyPlan p = plan_skinny(n, k);
  return p.splits > 1 ? (int64_t)p.splits * m * n * (int64_t)sizeof(float) : 0;
}

int mi355_linear(void* y, const void* x, const void* w, const void* bias, int64_t m, int64_t n,
                 int64_t k, int64_t x_stride, void* workspace, int64_t workspace_bytes,
                 int dtype, int out_dtype, void* stream) {
  MI355_REQUIRE(y && x && w, MI355_ERR_INVALID, "linear: null pointer");
  MI355_REQUIRE(m >= 0 && n > 0 && k > 0 && x_stride >= k, MI355_ERR_INVALID,
                "linear: bad shape m=%lld n=%lld k=%lld x_stride=%lld", (long long)m,
                (long long)n, (long long)k, (long long)x_stride);
  MI355_REQUIRE(out_dtype == dtype || out_dtype == MI355_F32, MI355_ERR_UNSUPPORTED,
                "linear: out_dtype must equal dtype or be fp32");
  MI355_REQUIRE(n < (1LL << 31) && k < (1LL << 31) && (n + 63) / 64 <= 0x7fffffffLL &&
                    (m + 63) / 64 <= 65535,
                MI355_ERR_UNSUPPORTED, "linear: shape too large");
  if (m == 0) return MI355_OK;
  bool skinny = skinny_ok(m, n, k, x_stride, x, w, dtype);
  if (skinny) {
    const int64_t need = mi355_linear_workspace_bytes(m, n, k);
    if (need > 0 && (!workspace || workspace_bytes < need || !aligned16(workspace))) {
      set_error("linear: workspace of %lld bytes (16-B aligned) required, got %lld",
                (long long)need, (long long)workspace_bytes);
      return MI355_ERR_INVALID;
    }
  }
  const bool out_f32 = out_dtype == MI355_F32 && dtype != MI355_F32;
  MI355_DISPATCH_DTYPE(dtype, "linear",
                       launch_linear<T>(y, x, w, bias, m, n, k, x_stride, workspace, skinny,
                                        out_f32, (hipStream_t)stream));
  MI355_CHECK_LAUNCH("linear");
  return MI355_OK;
}

}  // extern "C"
