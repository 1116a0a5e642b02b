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
#ifdef MI355_DEV_BUILD
#include "../../include/mi355_hotpath_dev.h"
#endif
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace mi355 {

// Epilogue store of one value with optional bias; OUT is T or float (float = T-rounded).
template <typename T, typename OUT> __device__ __forceinline__ OUT finish(float acc, float bias) {
  return (OUT)round_to<T>(acc + bias);
}

constexpr int SK_BK = 256;            // K elements per chunk
constexpr int SK_LDS_ROW = SK_BK + 16; // padded LDS row (elements): 544 B rows (bank = 8 * row: no ds_read_b128 conflicts)

// grid: (ceil(N / (64*NT)), splits).  block: 256.
template <typename T, int MT, int NT, typename OUT, bool PACKED = false>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ bias,
    OUT* __restrict__ y, float* __restrict__ slab, int M, int N, int K, int64_t x_stride,
    int k_per_split) {
  using frag = typename Mfma<T>::frag;
  constexpr int KMUL = PACKED ? 64 : 1;       // pointer advance per k element
  constexpr int R = 16 * MT;                  // activation rows held in LDS
  constexpr int XP = R * (SK_BK / 8) / 256;   // 16-B pieces of the x tile per thread (R/8)
  static_assert(XP >= 1, "tile too small");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* xs = reinterpret_cast<T*>(smem_raw);     // [2][R][SK_LDS_ROW]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  // 16-row weight tile of (wave, nt): the four waves of a workgroup own four adjacent tiles, so
  // one instruction of the workgroup reads 4 KiB that are contiguous in the packed layout.
  const int tile0 = blockIdx.x * (4 * NT) + wave;   // tile(nt) = tile0 + 4 * nt
  const int ngroups = (N + 63) >> 6;
  const int ks = blockIdx.y * k_per_split;
  const int ke = min(K, ks + k_per_split);
  const int nchunks = (ke - ks + SK_BK - 1) / SK_BK;

  // weight row pointers (clamped: rows >= N are loaded from row N-1 and never stored)
  const T* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int n = (tile0 + 4 * nt) * 16 + r16;
    n = n < N ? n : N - 1;
    const int grp = min(blockIdx.x * NT + nt, ngroups - 1);
    // packed: [N/64][K/32][4 tiles][64 lanes][8]
    wrow[nt] = PACKED ? w + (((int64_t)grp * (K >> 5) * 4 + wave) * 64 + lane) * 8
                      : w + (int64_t)n * K + q * 8;
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
          dst[nt][s] = *reinterpret_cast<const frag*>(wrow[nt] + (int64_t)(kc + s * 32) * KMUL);
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
    const int n0 = (tile0 + 4 * nt) * 16 + q * 4;
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


// ------------------------------------------------------------------------------------------
// Ring variant (K % 256 == 0): the decode-time kernel.
//   * chunk = 256/NT k-elements; every wave keeps THREE chunks of its weight rows in VGPRs
//     (computing / landed / in flight).  Queue order per iteration is x(c+1), W(c+2): the wait
//     for the activation tile leaves W(c+2) in flight, so the HBM stream never drains at the
//     per-chunk barrier.  All loads in the steady-state loop are unconditional, which lets the
//     compiler emit counted s_waitcnt vmcnt(N) instead of vmcnt(0).
//   * split-K partials are written in MFMA-fragment order (1-KiB coalesced stores) and the
//     LAST workgroup to finish an output block sums all partials in split order and writes y:
//     no second launch, deterministic, no spinning (the arrival counter is re-zeroed by the
//     workgroup that consumes it).
// grid: (ceil(N / (64*NT)), splits).  block: 256.
// ------------------------------------------------------------------------------------------
// Tuning / ablation switches.  They exist only in the development build of the library
// (-DMI355_DEV_BUILD, libmi355_hotpath_dev.so, entry point mi355_debug_set); in the product build
// they are compile-time zeros, so the library carries no process-global mutable state.
#ifdef MI355_DEV_BUILD
extern int g_decode_variant;
extern int g_extend_variant;
extern int g_mla_teams;
int g_gemm_variant = 0;
int g_force_splits = 0;
#else
static constexpr int g_gemm_variant = 0;
static constexpr int g_force_splits = 0;
#endif

template <typename T, int MT, int NT, typename OUT, bool PACKED, int WAVES = 4, bool ACT = false>
__global__ __launch_bounds__(64 * WAVES) void gemm_ring_kernel(
    const T* __restrict__ x, const T* __restrict__ w, const void* __restrict__ bias_,
    OUT* __restrict__ y, float* __restrict__ slab, unsigned* __restrict__ counters, int M, int N,
    int K, int64_t x_stride, int k_per_split, const float* __restrict__ x_scale,
    const float* __restrict__ w_scale) {
  using frag = typename Mfma<T>::frag;
  constexpr bool FP8 = std::is_same<T, fp8_t>::value;
  constexpr int VE = 16 / sizeof(T);          // elements per 16-byte fragment (8, or 16 for fp8)
  constexpr int KSTEP = 4 * VE;               // k covered by one fragment row (4 lane groups)
  constexpr int BK = 8 * KSTEP / NT;          // k elements per chunk (256 / NT; fp8: 512 / NT)
  constexpr int S = BK / KSTEP;               // fragment steps per chunk
  constexpr int ROW = BK + 2 * VE;            // padded LDS row (+32 B): bank = 8 * row (mod 64)
  constexpr int KMUL = PACKED ? 64 : 1;
  constexpr int R = 16 * MT;
  constexpr int THREADS = 64 * WAVES;
  constexpr int PIECES = R * (BK / VE);        // 16-B pieces of the x tile
  constexpr int XP = (PIECES + THREADS - 1) / THREADS;   // per thread (the last ones may repeat a piece)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* xs = reinterpret_cast<T*>(smem_raw);     // [2][R][ROW]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  // 16-row weight tile of (wave, nt): tile(nt) = tile0 + WAVES * nt.  WAVES tiles side by side per
  // workgroup share one activation tile; with WAVES = 4 they are one 64-row group of the packed
  // layout (4 KiB contiguous per instruction of the workgroup).
  const int tile0 = blockIdx.x * (WAVES * NT) + wave;
  const int ntiles = ((N + 63) >> 6) * 4;     // tiles that exist in the packed image
  const int ks = blockIdx.y * k_per_split;
  const int ke = min(K, ks + k_per_split);
  const int nchunks = (ke - ks) / BK;               // host guarantees divisibility, >= 1

  const T* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int tile = min(tile0 + WAVES * nt, ntiles - 1);
    int n = tile * 16 + r16;
    n = n < N ? n : N - 1;
    wrow[nt] = PACKED ? w + (((int64_t)(tile >> 2) * (K / KSTEP) * 4 + (tile & 3)) * 64 + lane) * VE + (int64_t)ks * KMUL
                      : w + (int64_t)n * K + ks + q * VE;
  }
  // x tile piece p -> (row, 16-B column)
  const T* xsrc[XP];
  int xdst[XP];
  bool xok[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    int p = threadIdx.x + i * THREADS;
    p = p < PIECES ? p : PIECES - 1;            // clamp, never predicate: the loads stay unconditional
    const int row = p / (BK / VE), c16 = p % (BK / VE);
    xok[i] = row < M;
    xsrc[i] = x + (int64_t)(row < M ? row : 0) * x_stride + ks + c16 * VE;
    xdst[i] = row * ROW + c16 * VE;
  }

  f32x4 acc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  frag w0[NT][S], w1[NT][S], w2[NT][S];
  Vec<T, VE> xreg[XP];

  auto load_w = [&](frag (&dst)[NT][S], int chunk) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const frag* src = reinterpret_cast<const frag*>(wrow[nt] + (int64_t)(chunk * BK + s * KSTEP) * KMUL);
        // packed weights are read exactly once, in whole lines: stream them past the caches
        dst[nt][s] = PACKED ? __builtin_nontemporal_load(src) : *src;
      }
  };
  auto load_x = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < XP; ++i) xreg[i] = ldvec<T, VE>(xsrc[i] + chunk * BK);
  };
  auto store_x = [&](int buf) {
    T* dst = xs + buf * (R * ROW);
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      if (!xok[i]) {
#pragma unroll
        for (int j = 0; j < VE; ++j) xreg[i].v[j] = T(0);
      }
      stvec<T, VE>(dst + xdst[i], xreg[i]);
    }
  };
  auto compute = [&](const frag (&wc)[NT][S], int buf) {
    const T* xt = xs + buf * (R * ROW) + r16 * ROW + q * VE;
    // The activation fragments of step s + 2 are requested before the MFMAs of step s issue, so
    // the LDS latency sits under matrix work instead of between every read and its MFMA.
    frag xf[3][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[0][mt] = *reinterpret_cast<const frag*>(xt + mt * 16 * ROW);
    if (S > 1) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[1][mt] = *reinterpret_cast<const frag*>(xt + mt * 16 * ROW + KSTEP);
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      if (s + 2 < S) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          xf[(s + 2) % 3][mt] = *reinterpret_cast<const frag*>(xt + mt * 16 * ROW + (s + 2) * KSTEP);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the matrix work (the scheduler would sink them)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = Mfma<T>::run(wc[nt][s], xf[s % 3][mt], acc[nt][mt]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto rotate = [&]() {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < S; ++s) {
        w0[nt][s] = w1[nt][s];
        w1[nt][s] = w2[nt][s];
      }
  };

  load_w(w0, 0);
  load_x(0);
  load_w(w1, nchunks > 1 ? 1 : 0);   // unconditional (clamped) so the wait below stays counted
  store_x(0);
  __syncthreads();

  int c = 0;
  for (; c + 2 < nchunks; ++c) {
    load_x(c + 1);
    load_w(w2, c + 2);
    compute(w0, c & 1);
    store_x((c + 1) & 1);
    __syncthreads();
    rotate();
  }
  if (c + 1 < nchunks) {
    load_x(c + 1);
    compute(w0, c & 1);
    store_x((c + 1) & 1);
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < S; ++s) w0[nt][s] = w1[nt][s];
    ++c;
  }
  compute(w0, c & 1);

  // Final value of one accumulator fragment: D[i = 16 tile + 4q + r][j = m (r16)].
  // ACT (SiLU-and-mul fused, weights packed with gate/up rows interleaved 8 + 8 per tile): rows
  // 0..7 of a tile are gate features 8 tile .. 8 tile + 7, rows 8..15 the matching up features, so
  // lane (m, q < 2) holds gate values whose up partners sit 32 lanes higher.
  // four consecutive output features of row m, stored as one 8 / 16-byte access when they all exist
  auto store4 = [&](OUT* row, int n, int limit, const OUT (&o)[4]) {
    if (n + 3 < limit && (limit & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
      Vec<OUT, 4> pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk.v[r] = o[r];
      stvec<OUT, 4>(row + n, pk);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < limit) row[n + r] = o[r];
    }
  };
  auto emit = [&](int nt, int mt, f32x4 v) {
    const int tile = tile0 + WAVES * nt;
    const int m = mt * 16 + r16;
    OUT o4[4];
    if constexpr (ACT && !FP8) {
      float mine[4], other[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        mine[r] = round_to<T>(v[r]);
        other[r] = __shfl_xor(mine[r], 32, 64);
      }
      // Lanes q < 2 hold gate values (their up partners 32 lanes higher), lanes q >= 2 the mirror image.
      // Every lane takes two of the four outputs of its pair -- the low lane r = 0, 1, the high lane r = 2, 3 --
      // so the activation's ~40 instructions per output run on all 64 lanes instead of 32; the high lane's two
      // results travel back as one packed word.  Same statement per output: same bits.
      const bool lo = q < 2;
      const float g0 = lo ? mine[0] : other[2], u0 = lo ? other[0] : mine[2];
      const float g1 = lo ? mine[1] : other[3], u1 = lo ? other[1] : mine[3];
      const T s0 = silu_mul_one<T>(from_f32<T>(g0), from_f32<T>(u0));
      const T s1 = silu_mul_one<T>(from_f32<T>(g1), from_f32<T>(u1));
      const Vec<T, 2> pk{{s0, s1}};
      const unsigned got = (unsigned)__shfl_xor((int)__builtin_bit_cast(unsigned, pk), 32, 64);
      if (lo && m < M) {
        const int half_n = N >> 1;
        const Vec<T, 2> hi = __builtin_bit_cast(Vec<T, 2>, got);
        o4[0] = (OUT)s0; o4[1] = (OUT)s1; o4[2] = (OUT)hi.v[0]; o4[3] = (OUT)hi.v[1];
        store4(y + (int64_t)m * half_n, tile * 8 + 4 * q, half_n, o4);
      }
    } else {
      if (m >= M) return;
      const int n0 = tile * 16 + q * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + r < N ? n0 + r : N - 1;
        if constexpr (FP8) {
          // apply_fp8_linear's epilogue: (acc * x_scale[m]) * w_scale[n] (+ bias), cast once
          float o = (v[r] * x_scale[m]) * w_scale[n];
          if (bias_) o += (float)static_cast<const OUT*>(bias_)[n];
          o4[r] = (OUT)o;
        } else {
          const T* bias = static_cast<const T*>(bias_);
          o4[r] = finish<T, OUT>(v[r], bias ? to_f32(bias[n]) : 0.f);
        }
      }
      store4(y + (int64_t)m * N, n0, N, o4);
    }
  };

  const int splits = gridDim.y;
  if constexpr (!ACT) {
    if (y == nullptr) {
      // Deferred reduction: this split's fp32 partial goes out row-major, [split][M][N], and the
      // CONSUMER kernel sums the splits (in split order, applies the fp8 scales if any, then rounds
      // once -- the same arithmetic as the combine below).  No ticket, no second pass here; the kernel boundary publishes the data.
      float* part = slab + (int64_t)blockIdx.y * M * N;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m = mt * 16 + r16, n0 = (tile0 + WAVES * nt) * 16 + q * 4;
          if (m >= M) continue;
          if (n0 + 3 < N && (N & 3) == 0) {
            *reinterpret_cast<f32x4*>(part + (int64_t)m * N + n0) = acc[nt][mt];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n0 + r < N) part[(int64_t)m * N + n0 + r] = acc[nt][mt][r];
          }
        }
      return;
    }
  }
  if (splits == 1) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) emit(nt, mt, acc[nt][mt]);
    return;
  }

  // split-K: partial tile in fragment order, [n-block][split][wave][nt][mt][lane][4], stored
  // write-through (sc1) so that no release fence is needed: once a wave's stores have drained
  // they are visible to every XCD.  The reducer reads them with sc1 loads (never a stale line).
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int TILE_BYTES = 16 * WAVES * NT * R * 4;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<char*>(slab) + (int64_t)blockIdx.x * splits * TILE_BYTES), 0,
      splits * TILE_BYTES, 0x00020000);
  const int frag_off = ((wave * NT * MT) * 256 + lane * 4) * 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[nt][mt]), rs,
                                             (int)blockIdx.y * TILE_BYTES + frag_off + (nt * MT + mt) * 1024,
                                             0, 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                    // all waves' stores drained; x tiles in LDS are dead
  int* last_flag = reinterpret_cast<int*>(smem_raw);
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(&counters[blockIdx.x], 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = (old == (unsigned)(splits - 1));
  }
  __syncthreads();
  if (!*last_flag) return;
  // Sum in split order.  Four splits' loads are issued back to back without conditions (clamped
  // address, masked value) so the reducer pays one memory round trip per four splits.
  f32x4 red[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < splits; s0 += 4) {
    u32x4 v[4][NT][MT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = min(s0 + u, splits - 1);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          v[u][nt][mt] = __builtin_amdgcn_raw_buffer_load_b128(
              rs, s * TILE_BYTES + frag_off + (nt * MT + mt) * 1024, 0, 16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 f = __builtin_bit_cast(f32x4, v[u][nt][mt]);
          if (s0 + u < splits) red[nt][mt] += f;
        }
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) emit(nt, mt, red[nt][mt]);
  if (threadIdx.x == 0) counters[blockIdx.x] = 0u;   // leave the counter zero for the next launch
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


// Re-order an nn.Linear weight [N, K] into MFMA A-operand fragment order
// [ceil(N/64)][K/32][4 tiles][64 lanes][8]: lane (r = l & 15, q = l >> 4) of block
// (group, kstep, t) holds W[64*group + 16*t + r][32*kstep + 8*q .. +7].  Rows >= N are zero.
// Every wave-instruction of the decode GEMM then reads one contiguous 1-KiB block, the four
// waves of a workgroup read 4 KiB side by side, and a workgroup walks K in one contiguous run.
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(T* __restrict__ dst, const T* __restrict__ src,
                                                          int64_t N, int64_t K) {
  const int64_t ksteps = K >> 5;
  const int64_t nblocks = ((N + 63) >> 6) * ksteps * 4;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nblocks * 64; v += (int64_t)gridDim.x * 256) {
    const int64_t blk = v >> 6;
    const int l = (int)(v & 63);
    const int64_t t = blk & 3, gk = blk >> 2;
    const int64_t grp = gk / ksteps, kstep = gk - grp * ksteps;
    const int64_t n = grp * 64 + t * 16 + (l & 15), k = kstep * 32 + (l >> 4) * 8;
    Vec<T, 8> val;
    if (n < N) {
      val = ldvec<T, 8>(src + n * K + k);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) val.v[j] = (T)0.f;
    }
    stvec<T, 8>(dst + v * 8, val);
  }
}

// The arrival counters always occupy the same leading bytes of the workspace, whatever the shape,
// so partial sums of one call can never land where another call expects zeroed counters.
constexpr int64_t SK_COUNTER_BYTES = 4096;

struct SkinnyPlan {
  int nt, waves, splits, k_per_split;
  int64_t nblocks;
  int64_t counter_bytes;   // arrival counters at the head of the workspace (must start zeroed)
  int64_t slab_bytes;      // fp32 split-K partials behind them
};

#ifdef MI355_DEV_BUILD
int g_force_waves = 0;
#else
static constexpr int g_force_waves = 0;
#endif

// Decomposition of y[M, N] = x W^T over workgroups: `waves` 16-row weight tiles (x NT) side by
// side per workgroup share one activation tile, K is split `splits` ways.  Aim: one or two
// workgroups per CU (256 CUs) with equal work, >= 2 chunks each; splitting K costs fp32 partial
// traffic (splits * M * N * 8 bytes written + read), so wide outputs are never split.
// Most splits a reduction of k elements (kchunks chunks of SK_BK) is cut into: two chunks per split, except
// that a short reduction (under four chunks: the low-rank projections of MLA) may go down to one -- there the
// launch is a latency chain and the extra workgroups are what shortens it.
static int64_t max_splits_for(int64_t k, int64_t kchunks) {
  return k < 4 * SK_BK ? std::max<int64_t>(1, kchunks) : std::min<int64_t>(16, k / 512);
}

static SkinnyPlan plan_skinny(int64_t m, int64_t n, int64_t k, bool ring) {
  SkinnyPlan p;
  // Very wide outputs (lm_head): two tiles per wave and, on the ring kernel, eight waves -- 256
  // rows per workgroup, so the activation tile is staged half as often per weight byte.  Up to 16
  // rows of x there is little to stage and one tile per wave keeps more workgroups in flight.
  p.nt = n >= 64 * 1024 && (m > 16 || !ring) ? 2 : 1;
  p.waves = p.nt == 2 && ring ? 8 : 4;
  const int64_t tiles = (n + 15) / 16;
  const int64_t kchunks = (k + SK_BK - 1) / SK_BK;
  int64_t splits = 1;
  if (ring && p.nt == 1) {
    // Cost model fitted to MI355X measurements (microseconds): a CU streams one 16-row x 256-k
    // weight tile in ~0.30, staging one activation chunk costs ~0.35, a launch ~4.5, a split-K
    // combine ~3 + 0.25 per split; workgroups beyond 256 queue up behind each other on a CU.
    double best = 1e30;
    for (int w : {4, 6, 7, 8}) {
      if (g_force_waves > 0 && w != g_force_waves) continue;
      const int64_t nb = (tiles + w - 1) / w;
      const int64_t smax = max_splits_for(k, kchunks);
      for (int64_t s = 1; s <= smax; ++s) {
        if (s > 1 && nb > SK_COUNTER_BYTES / 4) break;
        const int64_t chunks = (kchunks + s - 1) / s;
        const int64_t real_s = (kchunks + chunks - 1) / chunks;
        if (real_s != s) continue;
        const int64_t per_cu = (nb * s + 255) / 256;
        const double cost = per_cu * (w * chunks * 0.30 + chunks * 0.35) + 4.5 + (s > 1 ? 3.0 + 0.25 * s : 0.0);
        if (cost < best - 1e-9) { best = cost; p.waves = w; splits = s; }
      }
    }
  }
  const int64_t per = 16 * p.waves * p.nt;
  p.nblocks = (n + per - 1) / per;
  if (!(ring && p.nt == 1) && p.nblocks < 384 && p.nblocks <= SK_COUNTER_BYTES / 4) {
    splits = (320 + p.nblocks - 1) / p.nblocks;
    if (k / splits > 2048) splits = std::min<int64_t>((k + 1791) / 1792, 16);
  }
  const int64_t max_splits = max_splits_for(k, kchunks);
  splits = std::min(splits, max_splits);
  splits = std::max<int64_t>(splits, 1);
  if (g_force_splits > 0) splits = std::min<int64_t>(g_force_splits, max_splits);
  int64_t kps = (k + splits - 1) / splits;
  kps = (kps + SK_BK - 1) / SK_BK * SK_BK;
  p.k_per_split = (int)kps;
  p.splits = (int)((k + kps - 1) / kps);
  p.counter_bytes = p.splits > 1 ? SK_COUNTER_BYTES : 0;
  const int64_t rows = (m + 15) / 16 * 16;
  p.slab_bytes = p.splits > 1 ? (int64_t)p.splits * p.nblocks * per * rows * (int64_t)sizeof(float) : 0;
  return p;
}

// Warm-up geometry for the packed GEMM y[m, n] over k16 16-bit-equivalent K elements (an fp8
// weight counts K / 2): the first `budget_bytes` of the weight stream, spread evenly over the
// workgroups the GEMM will launch.  False when that GEMM does not run on the ring kernel.
bool plan_weight_prefetch(const void* w_packed, int64_t m, int64_t n, int64_t k16, int64_t budget_bytes,
                          WeightPrefetch* out) {
  if (!w_packed || m < 1 || m > 64 || n < 1 || k16 < SK_BK || k16 % SK_BK != 0 || g_gemm_variant != 0 ||
      n >= 64 * 1024 || budget_bytes <= 0)
    return false;
  const SkinnyPlan p = plan_skinny(m, n, k16, true);
  // A split-K plan is bound by its latency chain (few chunks per workgroup, then the combine), not
  // by the stream: warming it was measured to cost more in the carrier kernel than it returns.
  if (p.nt != 1 || p.splits > 1 || p.nblocks > 4096) return false;
  out->base = static_cast<const char*>(w_packed);
  out->nblocks = (int)p.nblocks;
  out->splits = p.splits;
  out->tiles_per_wg = p.waves;
  out->ntiles = (int)((n + 63) / 64 * 4);
  out->ksteps_total = (int)(k16 / 32);
  out->ksteps_per_split = p.k_per_split / 32;
  const int64_t per_wg = budget_bytes / (p.nblocks * p.splits);
  out->pf_steps = (int)std::min<int64_t>(out->ksteps_per_split, per_wg / (1024 * p.waves));
  return out->pf_steps > 0;
}

static bool skinny_ok(int64_t m, int64_t n, int64_t k, int64_t x_stride, const void* x,
                      const void* w, int dtype) {
  return (dtype == MI355_BF16 || dtype == MI355_F16) && m >= 1 && m <= 64 && k % 8 == 0 &&
         x_stride % 8 == 0 && aligned16(x) && aligned16(w) && n >= 1 && k >= 8;
}

static bool use_ring(int64_t k) { return k % SK_BK == 0 && g_gemm_variant == 0; }

template <typename T, int MT, int NT, typename OUT, bool PACKED, int WAVES, bool ACT = false>
static void launch_ring(const T* x, const T* w, const void* bias, OUT* y, float* slab, unsigned* counters,
                        int M, int N, int K, int64_t x_stride, const SkinnyPlan& p, hipStream_t st,
                        const float* x_scale = nullptr, const float* w_scale = nullptr) {
  const dim3 grid((unsigned)p.nblocks, (unsigned)p.splits);
  const size_t lds = 2 * (16 * MT) * (512 / NT + 32);   // bytes: same tile geometry for 16-bit and fp8
  auto kern = gemm_ring_kernel<T, MT, NT, OUT, PACKED, WAVES, ACT>;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES), lds, st, x, w, bias, y, slab, counters, M, N, K,
                     x_stride, p.k_per_split, x_scale, w_scale);
}

template <typename T, int MT, int NT, typename OUT, bool PACKED, bool ACT = false>
static void launch_skinny_cfg(const T* x, const T* w, const T* bias, OUT* y, char* ws, int M,
                              int N, int K, int64_t x_stride, const SkinnyPlan& p,
                              hipStream_t st, float* part = nullptr) {
  unsigned* counters = p.splits > 1 ? reinterpret_cast<unsigned*>(ws) : nullptr;
  float* slab = p.splits > 1 ? reinterpret_cast<float*>(ws + p.counter_bytes) : nullptr;
  if (part) {   // deferred reduction (ring kernel only; the caller checked): y == nullptr selects it
    counters = nullptr;
    slab = part;
    y = nullptr;
  }
  if (use_ring(K) || ACT) {
    if constexpr (PACKED && NT == 1) {
      if (p.waves == 8) return launch_ring<T, MT, NT, OUT, PACKED, 8, ACT>(x, w, bias, y, slab, counters, M, N, K, x_stride, p, st);
      if (p.waves == 7) return launch_ring<T, MT, NT, OUT, PACKED, 7, ACT>(x, w, bias, y, slab, counters, M, N, K, x_stride, p, st);
      if (p.waves == 6) return launch_ring<T, MT, NT, OUT, PACKED, 6, ACT>(x, w, bias, y, slab, counters, M, N, K, x_stride, p, st);
      return launch_ring<T, MT, NT, OUT, PACKED, 4, ACT>(x, w, bias, y, slab, counters, M, N, K, x_stride, p, st);
    }
    if constexpr (!ACT && PACKED && NT == 2)
      if (p.waves == 8) return launch_ring<T, MT, NT, OUT, PACKED, 8>(x, w, bias, y, slab, counters, M, N, K, x_stride, p, st);
    if constexpr (!ACT)
      return launch_ring<T, MT, NT, OUT, PACKED, 4>(x, w, bias, y, slab, counters, M, N, K, x_stride, p, st);
  }
  if constexpr (ACT) return;   // callers check the preconditions (ring kernel, one tile per wave)
  else {
  const dim3 grid((unsigned)p.nblocks, (unsigned)p.splits);
  const size_t lds = 2 * (16 * MT) * SK_LDS_ROW * sizeof(T);
  hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, NT, OUT, PACKED>), grid, dim3(256), lds, st, x, w, bias, y,
                     slab, M, N, K, x_stride, p.k_per_split);
  if (p.splits > 1) {
    const int64_t mn = (int64_t)M * N;
    hipLaunchKernelGGL((splitk_reduce_kernel<T, OUT>), dim3((unsigned)((mn + 255) / 256)),
                       dim3(256), 0, st, slab, bias, y, mn, N, p.splits);
  }
  }
}

template <typename T, typename OUT, bool PACKED = false, bool ACT = false>
static void launch_skinny(const T* x, const T* w, const T* bias, OUT* y, void* ws, int M, int N,
                          int K, int64_t x_stride, hipStream_t st, float* part = nullptr) {
  const SkinnyPlan p = plan_skinny(M, N, K, use_ring(K) && PACKED);
  const int mt = (M + 15) / 16;
#define SK(MT_, NT_) launch_skinny_cfg<T, MT_, NT_, OUT, PACKED, ACT>(x, w, bias, y, (char*)ws, M, N, K, x_stride, p, st, part)
  if constexpr (ACT) {
    if (mt <= 1) SK(1, 1); else if (mt <= 2) SK(2, 1); else SK(4, 1);
  } else if (p.nt == 2) {
    if (mt <= 1) SK(1, 2); else if (mt <= 2) SK(2, 2); else SK(4, 2);
  } else {
    if (mt <= 1) SK(1, 1); else if (mt <= 2) SK(2, 1); else SK(4, 1);
  }
#undef SK
}

template <typename T, typename OUT>
static void launch_linear_t(void* y, const void* x, const void* w, const void* bias, int64_t m,
                            int64_t n, int64_t k, int64_t x_stride, void* ws, bool skinny,
                            hipStream_t st) {
  if constexpr (!std::is_same<T, float>::value) {
    if (skinny) {
      launch_skinny<T, OUT>((const T*)x, (const T*)w, (const T*)bias, (OUT*)y, ws, (int)m, (int)n,
                            (int)k, x_stride, st);
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

// ---- fp8 W8A8 on the ring kernel -------------------------------------------------------------
// The packed image of an fp8 weight is the packed image of the same bytes viewed as [N, K/2]
// 16-bit elements (the layout is defined on 16-byte fragments), so mi355_pack_weight serves both.
static SkinnyPlan plan_fp8(int64_t m, int64_t n, int64_t k) {
  SkinnyPlan p = plan_skinny(m, n, k / 2, true);   // same bytes per chunk as a 16-bit GEMM of K/2
  p.k_per_split *= 2;
  return p;
}

template <typename OUT, int MT>
static void launch_fp8_mt(const fp8_t* x, const fp8_t* w, const void* bias, OUT* y, char* ws, int M, int N,
                          int K, const SkinnyPlan& p, const float* xs, const float* wsc, hipStream_t st,
                          float* part = nullptr) {
  unsigned* counters = p.splits > 1 ? reinterpret_cast<unsigned*>(ws) : nullptr;
  float* slab = p.splits > 1 ? reinterpret_cast<float*>(ws + p.counter_bytes) : nullptr;
  if (part) {   // deferred reduction: raw fp32 accumulator sums out, scales applied by the consumer
    counters = nullptr;
    slab = part;
    y = nullptr;
  }
#define F8R(W_) launch_ring<fp8_t, MT, 1, OUT, true, W_>(x, w, bias, y, slab, counters, M, N, K, K, p, st, xs, wsc)
  if (p.waves == 8) F8R(8); else if (p.waves == 7) F8R(7); else if (p.waves == 6) F8R(6); else F8R(4);
#undef F8R
}

template <typename OUT>
static void launch_fp8(const fp8_t* x, const fp8_t* w, const void* bias, OUT* y, char* ws, int M, int N, int K,
                       const float* xs, const float* wsc, hipStream_t st, float* part = nullptr) {
  const SkinnyPlan p = plan_fp8(M, N, K);
  const int mt = (M + 15) / 16;
  if (mt <= 1) launch_fp8_mt<OUT, 1>(x, w, bias, y, ws, M, N, K, p, xs, wsc, st, part);
  else if (mt <= 2) launch_fp8_mt<OUT, 2>(x, w, bias, y, ws, M, N, K, p, xs, wsc, st, part);
  else launch_fp8_mt<OUT, 4>(x, w, bias, y, ws, M, N, K, p, xs, wsc, st, part);
}

}  // namespace mi355

using namespace mi355;

extern "C" {

#ifdef MI355_DEV_BUILD
void mi355_debug_set(int key, int value) {
  if (key == 0) g_gemm_variant = value;   // 0 = ring kernel (default), 1 = two-launch kernel
  if (key == 2) g_force_splits = value;   // > 0 overrides the split-K plan
  if (key == 4) g_force_waves = value;    // 4 | 6 | 7 | 8 overrides the tiles-per-workgroup plan
  if (key == 6) g_decode_variant = value; // decode attention: 0 = product, 1 = P.V on the vector ALU, 3 = K rows straight into fragments
  if (key == 8) g_extend_variant = value; // prefill attention: 0 = 32x32x16 tile (product), 1 = 16x16x32 tile
  if (key == 10) g_mla_teams = value;     // MLA decode rings: 0 = two KV splits per workgroup when the grid allows (product), 1 = one
}
#endif

int64_t mi355_linear_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m < 1 || m > 64 || n < 1 || k < 1) return 0;
  // the larger of the two layouts (packed / row-major weights plan differently)
  const SkinnyPlan a = plan_skinny(m, n, k, use_ring(k)), b = plan_skinny(m, n, k, false);
  return std::max(a.counter_bytes + a.slab_bytes, b.counter_bytes + b.slab_bytes);
}

int64_t mi355_packed_weight_elems(int64_t n, int64_t k) { return ((n + 63) / 64) * 64 * k; }

int mi355_pack_weight(void* dst, const void* src, int64_t n, int64_t k, int dtype, void* stream) {
  MI355_REQUIRE(dst && src, MI355_ERR_INVALID, "pack_weight: null pointer");
  MI355_REQUIRE(n > 0 && k > 0 && k % 32 == 0, MI355_ERR_INVALID,
                "pack_weight: K (%lld) must be a positive multiple of 32", (long long)k);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "pack_weight: only 16-bit weights are packed");
  MI355_REQUIRE(aligned16(dst) && aligned16(src), MI355_ERR_INVALID, "pack_weight: pointers must be 16-B aligned");
  const int64_t vecs = ((n + 63) / 64) * 4 * (k / 32) * 64;
  const int grid = (int)std::min<int64_t>((vecs + 255) / 256, 256 * 32);
  if (dtype == MI355_BF16)
    hipLaunchKernelGGL((pack_weight_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)dst, (const bf16_t*)src, n, k);
  else
    hipLaunchKernelGGL((pack_weight_kernel<f16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (f16_t*)dst, (const f16_t*)src, n, k);
  MI355_CHECK_LAUNCH("pack_weight");
  return MI355_OK;
}

int mi355_linear_packed(void* y, const void* x, const void* w_packed, const void* bias, int64_t m,
                        int64_t n, int64_t k, int64_t x_stride, void* workspace,
                        int64_t workspace_bytes, int dtype, int out_dtype, void* stream) {
  MI355_REQUIRE(y && x && w_packed, MI355_ERR_INVALID, "linear_packed: null pointer");
  MI355_REQUIRE(m >= 0 && m <= 64 && n > 0 && k > 0 && k % 32 == 0 && x_stride >= k && x_stride % 8 == 0,
                MI355_ERR_INVALID, "linear_packed: needs 0 <= M <= 64, K %% 32 == 0 (m=%lld n=%lld k=%lld)",
                (long long)m, (long long)n, (long long)k);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "linear_packed: only 16-bit dtypes");
  MI355_REQUIRE(out_dtype == dtype || out_dtype == MI355_F32, MI355_ERR_UNSUPPORTED,
                "linear_packed: out_dtype must equal dtype or be fp32");
  MI355_REQUIRE(aligned16(x) && aligned16(w_packed), MI355_ERR_INVALID, "linear_packed: misaligned pointer");
  if (m == 0) return MI355_OK;
  const int64_t need = mi355_linear_workspace_bytes(m, n, k);
  if (need > 0 && (!workspace || workspace_bytes < need || !aligned16(workspace))) {
    set_error("linear_packed: workspace of %lld bytes (16-B aligned) required, got %lld", (long long)need,
              (long long)workspace_bytes);
    return MI355_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool f32o = out_dtype == MI355_F32;
  if (dtype == MI355_BF16) {
    if (f32o) launch_skinny<bf16_t, float, true>((const bf16_t*)x, (const bf16_t*)w_packed, (const bf16_t*)bias, (float*)y, workspace, (int)m, (int)n, (int)k, x_stride, st);
    else launch_skinny<bf16_t, bf16_t, true>((const bf16_t*)x, (const bf16_t*)w_packed, (const bf16_t*)bias, (bf16_t*)y, workspace, (int)m, (int)n, (int)k, x_stride, st);
  } else {
    if (f32o) launch_skinny<f16_t, float, true>((const f16_t*)x, (const f16_t*)w_packed, (const f16_t*)bias, (float*)y, workspace, (int)m, (int)n, (int)k, x_stride, st);
    else launch_skinny<f16_t, f16_t, true>((const f16_t*)x, (const f16_t*)w_packed, (const f16_t*)bias, (f16_t*)y, workspace, (int)m, (int)n, (int)k, x_stride, st);
  }
  MI355_CHECK_LAUNCH("linear_packed");
  return MI355_OK;
}

int mi355_linear_packed_splits(int64_t m, int64_t n, int64_t k) {
  if (m < 1 || m > 64 || n < 1 || n >= 64 * 1024 || k < SK_BK || k % SK_BK != 0 || g_gemm_variant != 0) return 0;
  return plan_skinny(m, n, k, true).splits;
}

int mi355_linear_packed_partials(float* part, const void* x, const void* w_packed, int64_t m, int64_t n,
                                 int64_t k, int64_t x_stride, int dtype, void* stream) {
  MI355_REQUIRE(part && x && w_packed, MI355_ERR_INVALID, "linear_packed_partials: null pointer");
  MI355_REQUIRE(m >= 0 && m <= 64 && n > 0 && k > 0 && x_stride >= k && x_stride % 8 == 0, MI355_ERR_INVALID,
                "linear_packed_partials: needs 0 <= M <= 64 (m=%lld n=%lld k=%lld)", (long long)m, (long long)n,
                (long long)k);
  MI355_REQUIRE(mi355_linear_packed_splits(m > 0 ? m : 1, n, k) > 0, MI355_ERR_UNSUPPORTED,
                "linear_packed_partials: needs K %% 256 == 0 and N < 65536 (k=%lld n=%lld)", (long long)k, (long long)n);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "linear_packed_partials: only 16-bit dtypes");
  MI355_REQUIRE(aligned16(x) && aligned16(w_packed) && aligned16(part), MI355_ERR_INVALID,
                "linear_packed_partials: misaligned pointer");
  if (m == 0) return MI355_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI355_BF16)
    launch_skinny<bf16_t, bf16_t, true>((const bf16_t*)x, (const bf16_t*)w_packed, nullptr, nullptr, nullptr, (int)m, (int)n, (int)k, x_stride, st, part);
  else
    launch_skinny<f16_t, f16_t, true>((const f16_t*)x, (const f16_t*)w_packed, nullptr, nullptr, nullptr, (int)m, (int)n, (int)k, x_stride, st, part);
  MI355_CHECK_LAUNCH("linear_packed_partials");
  return MI355_OK;
}

int mi355_linear_packed_silu_mul(void* y, const void* x, const void* w_packed, int64_t m, int64_t n,
                                 int64_t k, int64_t x_stride, void* workspace,
                                 int64_t workspace_bytes, int dtype, void* stream) {
  MI355_REQUIRE(y && x && w_packed, MI355_ERR_INVALID, "linear_packed_silu_mul: null pointer");
  MI355_REQUIRE(m >= 0 && m <= 64 && n > 0 && n % 16 == 0 && k > 0 && x_stride >= k && x_stride % 8 == 0,
                MI355_ERR_INVALID, "linear_packed_silu_mul: needs 0 <= M <= 64, N %% 16 == 0 (m=%lld n=%lld k=%lld)",
                (long long)m, (long long)n, (long long)k);
  MI355_REQUIRE(k % SK_BK == 0 && n < 64 * 1024 && g_gemm_variant == 0, MI355_ERR_UNSUPPORTED,
                "linear_packed_silu_mul: needs K %% 256 == 0 and N < 65536 (k=%lld n=%lld)", (long long)k,
                (long long)n);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "linear_packed_silu_mul: only 16-bit dtypes");
  MI355_REQUIRE(aligned16(x) && aligned16(w_packed), MI355_ERR_INVALID,
                "linear_packed_silu_mul: misaligned pointer");
  if (m == 0) return MI355_OK;
  const int64_t need = mi355_linear_workspace_bytes(m, n, k);
  if (need > 0 && (!workspace || workspace_bytes < need || !aligned16(workspace))) {
    set_error("linear_packed_silu_mul: workspace of %lld bytes (16-B aligned) required, got %lld",
              (long long)need, (long long)workspace_bytes);
    return MI355_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI355_BF16)
    launch_skinny<bf16_t, bf16_t, true, true>((const bf16_t*)x, (const bf16_t*)w_packed, nullptr, (bf16_t*)y, workspace, (int)m, (int)n, (int)k, x_stride, st);
  else
    launch_skinny<f16_t, f16_t, true, true>((const f16_t*)x, (const f16_t*)w_packed, nullptr, (f16_t*)y, workspace, (int)m, (int)n, (int)k, x_stride, st);
  MI355_CHECK_LAUNCH("linear_packed_silu_mul");
  return MI355_OK;
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

int64_t mi355_fp8_packed_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m < 1 || m > 64 || n < 1 || k < 512) return 0;
  const SkinnyPlan p = plan_fp8(m, n, k);
  return p.counter_bytes + p.slab_bytes;
}

int mi355_fp8_scaled_mm_packed(void* y, const uint8_t* xq, const uint8_t* wq_packed, const float* x_scale,
                               const float* w_scale, const void* bias, int64_t m, int64_t n, int64_t k,
                               void* workspace, int64_t workspace_bytes, int out_dtype, void* stream) {
  MI355_REQUIRE(y && xq && wq_packed && x_scale && w_scale, MI355_ERR_INVALID, "fp8_scaled_mm_packed: null pointer");
  MI355_REQUIRE(m >= 0 && m <= 64 && n > 0 && k > 0, MI355_ERR_INVALID,
                "fp8_scaled_mm_packed: needs 0 <= M <= 64 (m=%lld n=%lld k=%lld)", (long long)m, (long long)n,
                (long long)k);
  MI355_REQUIRE(k % 512 == 0 && n < 64 * 1024, MI355_ERR_UNSUPPORTED,
                "fp8_scaled_mm_packed: needs K %% 512 == 0 and N < 65536 (k=%lld n=%lld)", (long long)k, (long long)n);
  MI355_REQUIRE(aligned16(xq) && aligned16(wq_packed), MI355_ERR_INVALID, "fp8_scaled_mm_packed: misaligned pointer");
  if (m == 0) return MI355_OK;
  const int64_t need = mi355_fp8_packed_workspace_bytes(m, n, k);
  if (need > 0 && (!workspace || workspace_bytes < need || !aligned16(workspace))) {
    set_error("fp8_scaled_mm_packed: workspace of %lld bytes (16-B aligned) required, got %lld", (long long)need,
              (long long)workspace_bytes);
    return MI355_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (out_dtype) {
    case MI355_BF16: launch_fp8<bf16_t>(xq, wq_packed, bias, (bf16_t*)y, (char*)workspace, (int)m, (int)n, (int)k, x_scale, w_scale, st); break;
    case MI355_F16: launch_fp8<f16_t>(xq, wq_packed, bias, (f16_t*)y, (char*)workspace, (int)m, (int)n, (int)k, x_scale, w_scale, st); break;
    case MI355_F32: launch_fp8<float>(xq, wq_packed, bias, (float*)y, (char*)workspace, (int)m, (int)n, (int)k, x_scale, w_scale, st); break;
    default: set_error("fp8_scaled_mm_packed: unsupported out dtype %d", out_dtype); return MI355_ERR_UNSUPPORTED;
  }
  MI355_CHECK_LAUNCH("fp8_scaled_mm_packed");
  return MI355_OK;
}

int mi355_fp8_packed_splits(int64_t m, int64_t n, int64_t k) {
  if (m < 1 || m > 64 || n < 1 || n >= 64 * 1024 || k < 512 || k % 512 != 0 || g_gemm_variant != 0) return 0;
  return plan_fp8(m, n, k).splits;
}

int mi355_fp8_packed_partials(float* part, const uint8_t* xq, const uint8_t* wq_packed, int64_t m, int64_t n,
                              int64_t k, void* stream) {
  MI355_REQUIRE(part && xq && wq_packed, MI355_ERR_INVALID, "fp8_packed_partials: null pointer");
  MI355_REQUIRE(m >= 0 && m <= 64 && n > 0 && k > 0, MI355_ERR_INVALID,
                "fp8_packed_partials: needs 0 <= M <= 64 (m=%lld n=%lld k=%lld)", (long long)m, (long long)n, (long long)k);
  MI355_REQUIRE(mi355_fp8_packed_splits(m > 0 ? m : 1, n, k) > 0, MI355_ERR_UNSUPPORTED,
                "fp8_packed_partials: needs K %% 512 == 0 and N < 65536 (k=%lld n=%lld)", (long long)k, (long long)n);
  MI355_REQUIRE(aligned16(xq) && aligned16(wq_packed) && aligned16(part), MI355_ERR_INVALID,
                "fp8_packed_partials: misaligned pointer");
  if (m == 0) return MI355_OK;
  launch_fp8<bf16_t>(xq, wq_packed, nullptr, nullptr, nullptr, (int)m, (int)n, (int)k, nullptr, nullptr,
                     (hipStream_t)stream, part);
  MI355_CHECK_LAUNCH("fp8_packed_partials");
  return MI355_OK;
}

}  // extern "C"
