// Shared device/host helpers for the MI355X (gfx950, CDNA4) hot-path kernels.
// wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mi355_hotpath.h"

namespace mi355 {

using bf16_t = __bf16;      // native bf16: cast to/from float rounds to nearest even, keeps NaN
using f16_t = _Float16;

template <typename T> struct DType;
template <> struct DType<float> { static constexpr int code = MI355_F32; };
template <> struct DType<bf16_t> { static constexpr int code = MI355_BF16; };
template <> struct DType<f16_t> { static constexpr int code = MI355_F16; };

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }
// "Round through T": what a torch elementwise op on dtype T does to an fp32 intermediate.
template <typename T> __device__ __forceinline__ float round_to(float x) { return (float)(T)x; }
template <> __device__ __forceinline__ float round_to<float>(float x) { return x; }
// bf16: done on the bits (round to nearest even, NaN kept).  A (float)(__bf16)x round trip in
// registers is NOT safe: clang's default bf16 "excess precision" mode may drop the rounding.
template <> __device__ __forceinline__ float round_to<bf16_t>(float x) {
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return x;
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}

// N contiguous elements moved with one memory instruction (N * sizeof(T) in {4,8,16}).
template <typename T, int N> struct alignas(sizeof(T) * N) Vec {
  T v[N];
};
template <typename T, int N> __device__ __forceinline__ Vec<T, N> ldvec(const T* p) {
  return *reinterpret_cast<const Vec<T, N>*>(p);
}
// The same with the non-temporal cache policy: for bytes that are read once and not again before they
// would have been evicted anyway (a step's sweep over the KV cache).
template <typename T, int N> __device__ __forceinline__ Vec<T, N> ldvec_nt(const T* p) {
  static_assert(sizeof(T) * N == 16, "16-byte vectors only");
  typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(Vec<T, N>, __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(p)));
}
__device__ __forceinline__ uint2 ld8_nt(const void* p) {
  typedef unsigned nt_u32x2 __attribute__((ext_vector_type(2)));
  const nt_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x2*>(p));
  return uint2{v.x, v.y};
}
template <typename T, int N> __device__ __forceinline__ void stvec(T* p, const Vec<T, N>& x) {
  *reinterpret_cast<Vec<T, N>*>(p) = x;
}
// Elements per 16-byte access.
template <typename T> struct V16 { static constexpr int n = 16 / sizeof(T); };

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float x, float* red) {
  x = wave_sum(x);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();  // protect `red` against a previous use
  if (lane == 0) red[w] = x;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];  // same order in every thread: identical result
  return t;
}

// ---- host side -------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define MI355_REQUIRE(cond, code, ...)      \
  do {                                      \
    if (!(cond)) {                          \
      ::mi355::set_error(__VA_ARGS__);      \
      return (code);                        \
    }                                       \
  } while (0)

// After a launch: report (never sync).
#define MI355_CHECK_LAUNCH(name)                                                   \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      ::mi355::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
      return MI355_ERR_LAUNCH;                                                     \
    }                                                                              \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
__host__ __device__ static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Dispatch a lambda templated on the element type.
#define MI355_DISPATCH_DTYPE(dtype, NAME, ...)                               \
  switch (dtype) {                                                           \
    case MI355_F32: { using T = float; __VA_ARGS__; break; }                 \
    case MI355_BF16: { using T = ::mi355::bf16_t; __VA_ARGS__; break; }      \
    case MI355_F16: { using T = ::mi355::f16_t; __VA_ARGS__; break; }        \
    default:                                                                 \
      ::mi355::set_error("%s: unsupported dtype code %d", NAME, (int)dtype); \
      return MI355_ERR_UNSUPPORTED;                                          \
  }

// Eight values -> eight OCP e4m3fn bytes: clamp(v * inv, +-448), round to nearest even (the
// per-token quantiser's arithmetic, python/sglang/srt/layers/quantization/fp8_kernel.py:149-152).
template <typename T> __device__ __forceinline__ uint2 quant8_fp8(const Vec<T, 8>& v, float inv) {
  unsigned out[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = fminf(fmaxf(to_f32(v.v[4 * h + j]) * inv, -448.0f), 448.0f);
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], pk, true);
    out[h] = (unsigned)pk;
  }
  return uint2{out[0], out[1]};
}
// Eight OCP e4m3fn bytes -> eight values of T (exact: every e4m3 value is a bf16 / f16 / f32 value).
template <typename T> __device__ __forceinline__ Vec<T, 8> cvt8_from_fp8(uint2 raw) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  Vec<T, 8> o;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int word = (int)(h ? raw.y : raw.x);
    const f2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(word, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(word, true);
    o.v[4 * h + 0] = from_f32<T>(lo[0]);
    o.v[4 * h + 1] = from_f32<T>(lo[1]);
    o.v[4 * h + 2] = from_f32<T>(hi[0]);
    o.v[4 * h + 3] = from_f32<T>(hi[1]);
  }
  return o;
}
// Row maximum over a 256-thread workgroup (4 waves); `red` is 4 floats of LDS.
__device__ __forceinline__ float block_max256(float x, float* red) {
  x = wave_max(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// Strides (elements) of a KV cache buffer of `heads` heads x `d` features over `num_slots` slots:
// [num_slots, heads, d], or [heads, num_slots, d] when the dtype argument carries MI355_KV_HEAD_MAJOR, or one half
// of [heads, num_slots, 2, d] (keys | values) when it also carries MI355_KV_INTERLEAVED.
inline void kv_strides(int dtype_flags, int64_t num_slots, int heads, int d, int64_t& slot_stride, int64_t& head_stride) {
  if (dtype_flags & MI355_KV_HEAD_MAJOR) {
    const int64_t row = (dtype_flags & MI355_KV_INTERLEAVED) ? 2 * (int64_t)d : d;
    slot_stride = row;
    head_stride = num_slots * row;
  } else {
    slot_stride = (int64_t)heads * d;
    head_stride = d;
  }
}

// ---- L2 warm-up of the next GEMM's packed weights --------------------------------------------
// A kernel that leaves HBM idle (row norms, RoPE: tens of KB of traffic) can carry extra
// workgroups that pull the head of the NEXT skinny GEMM's weight stream into L2.  L2 contents
// survive the kernel boundary, and workgroup p of the warm-up lands on the same XCD as workgroup
// p of the GEMM (both are dispatched round-robin from workgroup 0), so each XCD's L2 holds exactly
// the bytes its GEMM workgroups ask for first.  Geometry is the GEMM planner's own.
struct WeightPrefetch {
  const char* base;       // packed weight image ([64-row group][k step][4 tiles][64 lanes][16 B])
  int nblocks, splits;    // the GEMM's grid
  int tiles_per_wg;       // 16-row tiles per workgroup
  int ntiles;             // tiles in the image
  int ksteps_total;       // K / (elements per fragment row)
  int ksteps_per_split;
  int pf_steps;           // k steps warmed per workgroup (0 = nothing to do)
};
bool plan_weight_prefetch(const void* w_packed, int64_t m, int64_t n, int64_t k16, int64_t budget_bytes,
                          WeightPrefetch* out);

// Called by all 256 threads of warm-up workgroup p.  Loads only; the xor keeps them alive.
__device__ __forceinline__ void prefetch_weights(const WeightPrefetch& d, int p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bx = p % d.nblocks, by = p / d.nblocks;
  const int t0 = bx * d.tiles_per_wg, s0 = by * d.ksteps_per_split;
  const int pieces = d.tiles_per_wg * d.pf_steps;     // 1-KiB pieces: (k step, tile)
  unsigned acc = 0;
  constexpr int U = 8;
  for (int i0 = wave; i0 < pieces; i0 += 4 * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(i0 + 4 * u, pieces - 1);
      const int tile = min(t0 + i % d.tiles_per_wg, d.ntiles - 1);
      const int step = min(s0 + i / d.tiles_per_wg, d.ksteps_total - 1);
      const char* src = d.base + (((int64_t)(tile >> 2) * d.ksteps_total + step) * 4 + (tile & 3)) * 1024 + lane * 16;
      v[u] = *reinterpret_cast<const uint4*>(src);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  asm volatile("" ::"v"(acc));
}

// silu(a) * b with the rounding points of F.silu(a) * b on dtype T (layers/activation.py:55-58).
template <typename T> __device__ __forceinline__ T silu_mul_one(T a, T b) {
  const float x = to_f32(a);
  const float s = round_to<T>(x / (1.0f + expf(-x)));  // F.silu on dtype T rounds here
  return from_f32<T>(s * to_f32(b));
}

// round_to<T> on two values at once.  For bf16 this is one v_cvt_pk_bf16_f32 (round to nearest even, the
// rounding round_to<bf16_t> does on the bits) and two bit operations; going through the integer view keeps
// the compiler from treating the round trip as a no-op.
typedef float rt_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rt_bf16x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ void round2_to(float a, float b, float& ra, float& rb) {
  ra = round_to<T>(a);
  rb = round_to<T>(b);
}
template <> __device__ __forceinline__ void round2_to<bf16_t>(float a, float b, float& ra, float& rb) {
  const rt_bf16x2 h = __builtin_convertvector(rt_f32x2{a, b}, rt_bf16x2);
  const uint32_t u = __builtin_bit_cast(uint32_t, h);
  ra = __uint_as_float(u << 16);
  rb = __uint_as_float(u & 0xffff0000u);
}

// v_mfma_f32_16x16x32 on 16-bit operands.  Lane l holds A[i = l & 15][k = 8 * (l >> 4) + 0..7] and
// B[k = 8 * (l >> 4) + 0..7][j = l & 15]; the result lane holds D[i = 4 * (l >> 4) + r][j = l & 15].
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
// v_mfma_f32_16x16x16 on 16-bit operands: lane l holds A[i = l & 15][k = 4 * (l >> 4) + 0..3] and
// B[k = 4 * (l >> 4) + 0..3][j = l & 15]; the result layout is that of the k = 32 instruction.
template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_t> {
  typedef short frag __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma16<f16_t> {
  typedef _Float16 frag __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  }
};

// v_mfma_f32_32x32x16 on 16-bit operands: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 8 h + 0..7]
// and B[k = 8 h + 0..7][col r]; result register i of lane l is D[row (i & 3) + 8 (i >> 2) + 4 h][col r].
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <typename T> struct Mfma32;
template <> struct Mfma32<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma32<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// fp8 (OCP e4m3fn): a 16-byte fragment feeds TWO k = 32 steps -- within a 64-wide K block lane
// group c owns bytes [16c, 16c + 16) and step h uses its bytes [8h, 8h + 8); the other operand
// uses the same map, so products pair up.
typedef unsigned char fp8_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <> struct Mfma<fp8_t> {
  using frag = i32x4;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    const long a0 = (long)(((unsigned long)(unsigned)a[1] << 32) | (unsigned)a[0]);
    const long a1 = (long)(((unsigned long)(unsigned)a[3] << 32) | (unsigned)a[2]);
    const long b0 = (long)(((unsigned long)(unsigned)b[1] << 32) | (unsigned)b[0]);
    const long b1 = (long)(((unsigned long)(unsigned)b[3] << 32) | (unsigned)b[2]);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, c, 0, 0, 0);
  }
};
template <> struct Mfma<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// One rotation pair with the rounding points of rope_kernel (cos, sin -> T; every product and the sum /
// difference round to T on their own).
template <typename T>
__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float sn, T& o1, T& o2) {
#pragma clang fp contract(off)
  c = round_to<T>(c);
  sn = round_to<T>(sn);
  o1 = from_f32<T>(round_to<T>(round_to<T>(x1 * c) - round_to<T>(x2 * sn)));
  o2 = from_f32<T>(round_to<T>(round_to<T>(x2 * c) + round_to<T>(x1 * sn)));
}
// ds_read_b64_tr_b16: per 16 lanes a block of 4 rows x 16 columns of 16-bit elements, delivered
// column-major -- lane i of the group receives column i of the four rows.  Lane 4 q + pp supplies
// the address of row q, columns 4 pp .. 4 pp + 3 (8-byte aligned).  EXEC must be all ones.
template <typename T> struct TrRead {
  typedef short s4 __attribute__((ext_vector_type(4)));
  typedef T v4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ v4 run(const T* lds) {
    const s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds));
    return __builtin_bit_cast(v4, r);
  }
};

}  // namespace mi355
