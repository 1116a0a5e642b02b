// Large-M projection GEMM  y[M, N] = x[M, K] @ W[N, K]^T  for gfx950, fp32 accumulation on the matrix
// cores: the prefill side of UnquantizedLinearMethod.apply
// (python/sglang/srt/layers/quantization/unquant.py:110-122) for 16-bit operands and of
// apply_fp8_linear (layers/quantization/fp8_utils.py:676-749) for fp8 W8A8 operands with per-token /
// per-channel scales, with the MLP's SiLU-and-mul (python/sglang/srt/models/llama.py:94-98,
// layers/activation.py:55-65) optionally applied in the epilogue.  The decode side (M <= 64) is the
// weight-streaming kernel in gemm.hip.
//
// Structure (one 512-thread workgroup per CU, 128 KiB of LDS, one 256 x 256 output tile):
//   * K is walked in tiles of 128 BYTES per row (64 16-bit or 128 fp8 elements); a tile's operands
//     live in LDS as four HALF-TILES of 128 rows x 128 bytes (x rows 0..127, x rows 128..255, W rows
//     0..127, W rows 128..255), two tiles resident;
//   * the load unit writes LDS itself (global_load_lds_dwordx4, 8 rows x 128 B per wave-instruction:
//     whole cache lines);
//   * the eight waves form two groups of four (group = row half of the tile, wave-in-group = 64
//     output columns).  A wave alternates a READ slot (its 8 load requests, then 24 ds_read_b128: its A
//     and B fragments of a whole K tile) with a MATRIX slot (nothing but 64 x v_mfma_f32_16x16x32 on
//     16-bit operands, or 32 x v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales on fp8
//     operands: twice the 16-bit rate per byte staged); the groups run one slot apart, so on every
//     SIMD one wave's matrix work covers the other wave's LDS reads and load issue (a load-to-LDS
//     holds the issuing wave for ~40 cycles: anywhere in a matrix slot that time is lost to the
//     matrix pipe).  One s_barrier per slot;
//   * a half-tile is requested by the group that owns it, one to two tiles ahead, as soon as its LDS
//     region is free, and is waited for (vmcnt(0)) by the requesting waves at the end of their next
//     MATRIX slot: at least one slot in flight, one barrier before its first reader;
//   * LDS rows are 128 bytes; the 16-byte chunk index is XOR-ed with a function of the row on the
//     SOURCE address of the load (the load unit writes lane-linear) and again on the fragment read:
//     conflict-free ds_read_b128 for 16-byte (16-bit) and 32-byte (fp8) fragments.
//   * W rows are assigned to MFMA row slots so that a lane ends up with 16 consecutive output
//     columns (or, for SiLU-and-mul, 8 gate columns and the 8 matching up columns): the epilogue
//     writes whole 128-byte lines with 16-byte stores and needs no cross-lane traffic.
//   * workgroup -> tile: consecutive workgroups round-robin over the 8 XCDs, so workgroups b and
//     b + 8 share an L2; each XCD works through 8 x 4 blocks of tiles (32 CUs = one block).
#include "common.cuh"

namespace mi355 {
namespace {

constexpr int GT_HALF = 128 * 128;      // bytes of one half-tile
constexpr int GT_STAGE = 4 * GT_HALF;   // one K tile: x (2 halves) + W (2 halves)
constexpr int GT_LDS = 2 * GT_STAGE;    // 131072

struct GemmTiledParams {
  const void* x;
  const void* w;
  const void* bias;
  void* y;
  const float* x_scale;   // fp8: per-token scales [M]
  const float* w_scale;   // fp8: per-channel scales [N]
  int M, N, K;      // N = rows of W
  int64_t ldx;      // elements between rows of x
  int ldy;          // elements between rows of y
  int mt, nt;       // output tiles along M and N
  int snt, nsuper;  // 8 x 4 blocks of tiles: count along N, total
  // EPI 2 (qkv projection + RoPE + KV-cache write)
  const int64_t* positions;   // [M]
  const void* cos_sin;        // [max_pos, 128]: cos | sin, fp32 or the tensor dtype
  int cs_f32;
  int hq, hkv;                // local query / kv heads (head_dim 128)
  void* k_buffer;             // [slots, hkv, 128], or head-major [hkv, slots, 128]
  void* v_buffer;
  int64_t kv_slot, kv_head;   // elements between consecutive slots / kv heads of the two buffers
  const void* loc;            // [M] cache slot of each row, int64 or int32
  int loc_i32;
};

__device__ __forceinline__ void glds16(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

typedef int i32x8 __attribute__((ext_vector_type(8)));

// XOR applied to the 16-byte chunk index of LDS row r (depends on r & 15 only).  16-bit fragments
// are one chunk per lane (chunk 4 kk + (lane >> 4)), fp8 fragments two (chunks 2 (lane >> 4) + {0, 1}):
// each needs its own function for the 16-lane service groups of ds_read_b128 to hit 16 distinct
// 16-byte bank groups.
template <bool FP8> __device__ __forceinline__ int gt_swz(int r) {
  return FP8 ? (((r >> 1) & 1) | (((r >> 3) & 1) << 2)) : ((r >> 1) & 7);
}

// T: output dtype and, for 16-bit GEMMs, the operand dtype.  FP8: operands are OCP e4m3fn bytes.
// EPI 0: y = T(acc (+ bias)); EPI 1: W = [gate; up] (N = 2 I rows), y[m, j] = silu_mul(T(gate_j), T(up_j));
// fp8: acc is first scaled, (acc * x_scale[m]) * w_scale[n].
// Epilogue of the qkv projection (EPI 2): neox RoPE over the whole 128-wide head on the q and k heads with
// the rounding points of mi355_rope_inplace (cos / sin -> T; every product and sum rounds to T on its own, so
// no contraction here); rotated q / k and plain v go to y, the k and v rows also to the KV cache.
// The lane holds, for each of its 8 rows, columns d0 .. d0 + 7 of one head (acc[mi][0..1]) and their rotation
// partners d0 + 64 .. d0 + 71 (acc[mi][2..3]).  Every load (positions, cache slots, cos / sin rows) is
// requested before the arithmetic of the rows it belongs to: nothing else runs on the CU during an epilogue,
// so a chain of dependent loads per row would be paid in full.
template <typename T, bool FP8, bool CSF32>
__device__ __forceinline__ void qkv_rope_epilogue(const GemmTiledParams& p, f32x4 (&acc)[8][4], const float (&sxm)[8],
                                                  int tm, int tn, int g, int wc, int q4, int r16) {
#pragma clang fp contract(off)
  T* yg = (T*)p.y;
  const int head = 2 * tn + (wc >> 1), d0 = (wc & 1) * 32 + q4 * 8;
  const int col0 = head * 128 + d0;
  const int mbase = tm * 256 + g * 128 + r16;
  const bool rot = head < p.hq + p.hkv;
  T* cache_dst = nullptr;
  if (head >= p.hq + p.hkv) cache_dst = (T*)p.v_buffer + (int64_t)(head - p.hq - p.hkv) * p.kv_head + d0;
  else if (head >= p.hq) cache_dst = (T*)p.k_buffer + (int64_t)(head - p.hq) * p.kv_head + d0;
  float slo[8], shi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    slo[j] = FP8 ? p.w_scale[col0 + j] : 1.f;
    shi[j] = FP8 ? p.w_scale[col0 + 64 + j] : 1.f;
  }
  int posv[8], slotv[8];   // positions and slot numbers fit 31 bits (the host checks the slot count)
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = mbase + mi * 16;
    posv[mi] = (int)p.positions[m < p.M ? m : p.M - 1];
    slotv[mi] = 0;
  }
  if (cache_dst != nullptr) {
    if (p.loc_i32) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int m = mbase + mi * 16;
        slotv[mi] = ((const int32_t*)p.loc)[m < p.M ? m : p.M - 1];
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int m = mbase + mi * 16;
        slotv[mi] = (int)((const int64_t*)p.loc)[m < p.M ? m : p.M - 1];
      }
    }
  }
  constexpr int G = (CSF32 || FP8) ? 2 : 4;   // rows whose cos / sin are in flight together (register budget)
#pragma unroll
  for (int half = 0; half < 8 / G; ++half) {
    // cos / sin of G rows (v heads read them too and ignore them: the branch below is wave-uniform)
    float c[G][8], sn[G][8];
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int64_t pos = posv[half * G + i];
      if constexpr (CSF32) {
        const float* cs = (const float*)p.cos_sin + pos * 128 + d0;
        const float4 c0 = *reinterpret_cast<const float4*>(cs), c1 = *reinterpret_cast<const float4*>(cs + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(cs + 64), s1 = *reinterpret_cast<const float4*>(cs + 68);
        c[i][0] = c0.x; c[i][1] = c0.y; c[i][2] = c0.z; c[i][3] = c0.w;
        c[i][4] = c1.x; c[i][5] = c1.y; c[i][6] = c1.z; c[i][7] = c1.w;
        sn[i][0] = s0.x; sn[i][1] = s0.y; sn[i][2] = s0.z; sn[i][3] = s0.w;
        sn[i][4] = s1.x; sn[i][5] = s1.y; sn[i][6] = s1.z; sn[i][7] = s1.w;
      } else {
        const T* cs = (const T*)p.cos_sin + pos * 128 + d0;
        const Vec<T, 8> cv = ldvec<T, 8>(cs), sv = ldvec<T, 8>(cs + 64);
#pragma unroll
        for (int j = 0; j < 8; ++j) { c[i][j] = to_f32(cv.v[j]); sn[i][j] = to_f32(sv.v[j]); }
      }
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int mi = half * G + i;
      const int m = mbase + mi * 16;
      float x1[8], x2[8];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float lo = acc[mi][h][r], hi = acc[mi][2 + h][r];
          if (FP8) { lo = (lo * sxm[mi]) * slo[h * 4 + r]; hi = (hi * sxm[mi]) * shi[h * 4 + r]; }
          round2_to<T>(lo, hi, x1[h * 4 + r], x2[h * 4 + r]);   // the projection's own result, rounded to T
        }
      Vec<T, 8> o1, o2;
      if (rot) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float cj = c[i][j], sj = sn[i][j];
          if constexpr (CSF32) round2_to<T>(cj, sj, cj, sj);   // a table of dtype T holds rounded values already
          float a1, a2, b1, b2;
          round2_to<T>(x1[j] * cj, x2[j] * sj, a1, a2);
          round2_to<T>(x2[j] * cj, x1[j] * sj, b1, b2);
          o1.v[j] = from_f32<T>(a1 - a2);
          o2.v[j] = from_f32<T>(b1 + b2);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { o1.v[j] = from_f32<T>(x1[j]); o2.v[j] = from_f32<T>(x2[j]); }
      }
      if (m < p.M) {
        stvec<T, 8>(yg + (int64_t)m * p.ldy + col0, o1);
        stvec<T, 8>(yg + (int64_t)m * p.ldy + col0 + 64, o2);
        if (cache_dst != nullptr) {
          T* dst = cache_dst + (int64_t)slotv[mi] * p.kv_slot;
          stvec<T, 8>(dst, o1);
          stvec<T, 8>(dst + 64, o2);
        }
      }
    }
  }
}

template <typename T, int EPI, bool FP8>
__global__ __launch_bounds__(512) void gemm_tiled_kernel(GemmTiledParams p) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  using frag = typename Mfma<T>::frag;
  constexpr int ES = FP8 ? 1 : 2;   // operand element bytes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;

  // ---- workgroup -> output tile ------------------------------------------------------------
  const int b = blockIdx.x, li = b >> 3;
  const int st = (li >> 5) * 8 + (b & 7), within = li & 31;
  if (st >= p.nsuper) return;
  const int sup_m = st / p.snt, sup_n = st - sup_m * p.snt;
  const int tm = sup_m * 8 + (within & 7), tn = sup_n * 4 + (within >> 3);
  if (tm >= p.mt || tn >= p.nt) return;

  const int T_ = (p.K * ES) >> 7;   // K tiles of 128 bytes per row
  const char* xg = (const char*)p.x;
  const char* wg = (const char*)p.w;
  const int ncols = EPI == 1 ? (p.N >> 1) : p.N;   // output columns

  // ---- per-lane source pointers of the eight load instructions a wave issues per K tile -----
  // A wave requests data only in its READ slot (a load-to-LDS holds the issuing wave ~40 cycles: in
  // front of or between the MFMAs of a matrix slot that time is lost to the matrix pipe, next to the
  // LDS reads of a READ slot it is covered by the other group's MFMAs).  Group 0 (READ slot 2t) requests
  // W of tile t + 1, both halves; group 1 (READ slot 2t + 1) requests x rows 128..255 of tile t + 1 and
  // x rows 0..127 of tile t + 2: each is free when requested (its last reader finished a slot earlier)
  // and is waited for -- by the waves that requested it, at the end of their MATRIX slot -- before
  // the barrier in front of the first slot that reads it.  A half-tile is 16 pieces of 8 rows x 128
  // bytes; wave wc of the group requests pieces 4 wc .. 4 wc + 3 of its group's two half-tiles:
  //   ldA: group 0 W rows 0..127 (tile t + 1), group 1 x rows 128..255 (tile t + 1)
  //   ldB: group 0 W rows 128..255 (tile t + 1), group 1 x rows 0..127 (tile t + 2)
  const char* ldA[4];
  const char* ldB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = (wc * 4 + i) * 8 + (lane >> 3);             // row inside the half-tile
    const int c = (lane & 7) ^ gt_swz<FP8>(rr);                // logical 16-byte chunk this lane fetches
    const char* xs[2];
    const char* ws[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int m = tm * 256 + h * 128 + rr;
      m = m < p.M ? m : p.M - 1;
      xs[h] = xg + ((int64_t)m * p.ldx) * ES + c * 16;
      const int rb = h * 128 + rr, wcb = rb >> 6, sl = rb & 63, ni = sl >> 4, ii = sl & 15;
      int n;
      if (EPI == 1) {
        int j = tn * 128 + wcb * 32 + (ii >> 2) * 8 + (ni & 1) * 4 + (ii & 3);
        j = j < ncols ? j : ncols - 1;
        n = j + (ni >> 1) * ncols;
      } else if (EPI == 2) {
        // a tile is two heads; wave wcb serves head 2 tn + (wcb >> 1), rotation pairs d and d + 64 with
        // d = 32 (wcb & 1) + 8 (ii >> 2) + 4 (ni & 1) + (ii & 3): a lane ends with 8 d's and their partners
        n = (2 * tn + (wcb >> 1)) * 128 + (wcb & 1) * 32 + (ii >> 2) * 8 + (ni & 1) * 4 + (ii & 3) + (ni >> 1) * 64;
      } else {
        n = tn * 256 + wcb * 64 + (ii >> 2) * 16 + ni * 4 + (ii & 3);
        n = n < p.N ? n : p.N - 1;
      }
      ws[h] = wg + ((int64_t)n * p.K) * ES + c * 16;
    }
    ldA[i] = g == 0 ? ws[0] : xs[1];
    ldB[i] = g == 0 ? ws[1] : xs[0];
  }
  // LDS image of a K tile: x rows 0..255 (two halves), then W rows 0..255 (two halves)
  constexpr int LX0 = 0, LX1 = GT_HALF, LW0 = 2 * GT_HALF, LW1 = 3 * GT_HALF;
  char* const lds_w = sm + wc * 4096;   // this wave's four 1-KiB pieces inside a half-tile
  const int ldA_l = g == 0 ? LW0 : LX1, ldB_l = g == 0 ? LW1 : LX0;
  const int ldB_d = g == 0 ? 1 : 2;     // tile distance of the second request (the first is always t + 1)

  // ---- prologue: all of tile 0 and x rows 0..127 of tile 1 ------------------------------------
#pragma unroll
  for (int i = 0; i < 4; ++i) glds16(ldA[i], lds_w + ldA_l + i * 1024);            // W0(0) | x1(0)
#pragma unroll
  for (int i = 0; i < 4; ++i) glds16(ldB[i], lds_w + ldB_l + i * 1024);            // W1(0) | x0(0)
  if (g == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(ldB[i] + 128, lds_w + GT_STAGE + ldB_l + i * 1024);   // x0(1)
  }

  // ---- fragment read addresses ----------------------------------------------------------------
  // 16-bit: fragment of k-step kk = chunk 4 kk + q4 (k-step 1 is rd0 ^ 64); fp8: the fragment of the
  // whole 128-deep tile = chunks 2 q4 and 2 q4 + 1 (rd0 and rd0 ^ 16)
  const int r16 = lane & 15, q4 = lane >> 4;
  const int rd0 = r16 * 128 + (((FP8 ? 2 * q4 : q4) ^ gt_swz<FP8>(r16)) << 4);
  const int a_off = g * GT_HALF, b_off = 2 * GT_HALF + wc * 8192;

  f32x4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  // two 16-byte registers per fragment slot: k-steps 0 / 1 (16-bit) or the two halves of the 32-byte
  // fp8 fragment
  i32x4 af[2][8], bf[2][4];
#ifndef GT_ABL
#define GT_ABL 0   // development ablations: 1 no loads in the loop, 2 no fragment reads, 3 no MFMA, 4 no vmcnt waits
#endif
  if (GT_ABL == 2) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) af[kk][mi] = *reinterpret_cast<const i32x4*>(sm + rd0 + mi * 2048 + kk * 64);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf[kk][ni] = *reinterpret_cast<const i32x4*>(sm + rd0 + ni * 2048 + 32768 + kk * 64);
    }
  }

// four 1-KiB pieces of one half-tile of tile `tl`; TAIL: nothing past the last tile
#define GT_LOAD4(ptr, ldso, tl, TAIL)                                                        \
  do {                                                                                       \
    if (GT_ABL != 1 && (!(TAIL) || (tl) < T_)) {                                             \
      char* d_ = lds_w + (((tl) & 1) ? GT_STAGE : 0) + (ldso);                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                          \
          glds16(ptr[i] + (int64_t)(tl) * 128, d_ + i * 1024);                               \
    }                                                                                        \
  } while (0)
#define GT_SLOT_END()                                                                        \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    __builtin_amdgcn_s_barrier();                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
#define GT_TILE(t, TAIL)                                                                     \
  do {                                                                                       \
    const int cur_ = ((t) & 1) ? GT_STAGE : 0;                                               \
    /* READ slot: request the group's two half-tiles, read the fragments of the whole K tile t */ \
    GT_LOAD4(ldA, ldA_l, (t) + 1, TAIL);                                                     \
    GT_LOAD4(ldB, ldB_l, (t) + ldB_d, TAIL);                                                 \
    if (GT_ABL != 2) {                                                                       \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                     \
        const int ro_ = rd0 ^ (FP8 ? (kk << 4) : (kk << 6));                                 \
        const char* ab_ = sm + cur_ + a_off + ro_;                                           \
        const char* bb_ = sm + cur_ + b_off + ro_;                                           \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                     \
            bf[kk][ni] = *reinterpret_cast<const i32x4*>(bb_ + ni * 2048);                   \
        _Pragma("unroll") for (int mi = 0; mi < 8; ++mi)                                     \
            af[kk][mi] = *reinterpret_cast<const i32x4*>(ab_ + mi * 2048);                   \
      }                                                                                      \
    }                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
    GT_SLOT_END();                                                                           \
    /* MATRIX slot: nothing but MFMAs, then the wave's requests of its READ slot must have landed */ \
    if (GT_ABL == 3) {                                                                       \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                     \
        _Pragma("unroll") for (int mi = 0; mi < 8; ++mi) asm volatile("" ::"v"(af[kk][mi])); \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) asm volatile("" ::"v"(bf[kk][ni])); \
      }                                                                                      \
    } else {                                                                                 \
      __builtin_amdgcn_s_setprio(1);                                                         \
      if constexpr (FP8) {                                                                   \
        _Pragma("unroll") for (int mi = 0; mi < 8; ++mi)                                     \
          _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) {                                 \
            const i32x8 a8_ = __builtin_shufflevector(bf[0][ni], bf[1][ni], 0, 1, 2, 3, 4, 5, 6, 7); \
            const i32x8 b8_ = __builtin_shufflevector(af[0][mi], af[1][mi], 0, 1, 2, 3, 4, 5, 6, 7); \
            acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                  \
                a8_, b8_, acc[mi][ni], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);                  \
          }                                                                                  \
      } else {                                                                               \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                     \
          _Pragma("unroll") for (int mi = 0; mi < 8; ++mi)                                   \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                 \
                acc[mi][ni] = Mfma<T>::run(__builtin_bit_cast(frag, bf[kk][ni]),             \
                                           __builtin_bit_cast(frag, af[kk][mi]), acc[mi][ni]); \
      }                                                                                      \
      __builtin_amdgcn_s_setprio(0);                                                         \
    }                                                                                        \
    if (GT_ABL != 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
    GT_SLOT_END();                                                                           \
  } while (0)

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile 0 (and x rows 0..127 of tile 1) have landed
  GT_SLOT_END();
  if (g == 1) GT_SLOT_END();   // the second group runs one slot behind the first
  int t = 0;
  for (; t + 2 < T_; ++t) GT_TILE(t, false);   // every request exists: no conditions in the loop
  for (; t < T_; ++t) GT_TILE(t, true);
  if (g == 0) __builtin_amdgcn_s_barrier();
#undef GT_LOAD4
#undef GT_SLOT_END
#undef GT_TILE

  // ---- epilogue ---------------------------------------------------------------------------------
  // acc[mi][ni][r]: row m = 16 mi + (lane & 15) of the wave's 128 rows; W slot 16 ni + 4 q4 + r, which
  // the load mapping above made output column 16 q4 + 4 ni + r (plain) or gate/up column 8 q4 + 4 (ni & 1) + r.
  T* yg = (T*)p.y;
  const T* bias = (const T*)p.bias;
  float sxm[8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = tm * 256 + g * 128 + mi * 16 + r16;
    sxm[mi] = FP8 ? p.x_scale[m < p.M ? m : p.M - 1] : 1.f;
  }
  if constexpr (EPI == 2) {
    if (p.cs_f32) qkv_rope_epilogue<T, FP8, true>(p, acc, sxm, tm, tn, g, wc, q4, r16);
    else qkv_rope_epilogue<T, FP8, false>(p, acc, sxm, tm, tn, g, wc, q4, r16);
  } else if (EPI == 1) {
    const int j0 = tn * 128 + wc * 32 + q4 * 8;
    if (j0 < ncols) {
      float sg[8], su[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sg[j] = FP8 ? p.w_scale[j0 + j] : 1.f;
        su[j] = FP8 ? p.w_scale[ncols + j0 + j] : 1.f;
      }
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int m = tm * 256 + g * 128 + mi * 16 + r16;
        Vec<T, 8> o;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float gv = acc[mi][h][r], uv = acc[mi][2 + h][r];
            if (FP8) { gv = (gv * sxm[mi]) * sg[h * 4 + r]; uv = (uv * sxm[mi]) * su[h * 4 + r]; }
            o.v[h * 4 + r] = silu_mul_one<T>(from_f32<T>(gv), from_f32<T>(uv));
          }
        if (m < p.M) stvec<T, 8>(yg + (int64_t)m * p.ldy + j0, o);
      }
    }
  } else {
    const int n0 = tn * 256 + wc * 64 + q4 * 16;
    if (n0 < p.N) {
      float bv[16], sw[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) { bv[j] = 0.f; sw[j] = FP8 ? p.w_scale[n0 + j] : 1.f; }
      if (bias != nullptr) {
        const Vec<T, 8> b0 = ldvec<T, 8>(bias + n0), b1 = ldvec<T, 8>(bias + n0 + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { bv[j] = to_f32(b0.v[j]); bv[8 + j] = to_f32(b1.v[j]); }
      }
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int m = tm * 256 + g * 128 + mi * 16 + r16;
        Vec<T, 8> o0, o1;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v0 = acc[mi][ni][r], v1 = acc[mi][2 + ni][r];
            if (FP8) { v0 = (v0 * sxm[mi]) * sw[ni * 4 + r]; v1 = (v1 * sxm[mi]) * sw[8 + ni * 4 + r]; }
            o0.v[ni * 4 + r] = from_f32<T>(v0 + bv[ni * 4 + r]);
            o1.v[ni * 4 + r] = from_f32<T>(v1 + bv[8 + ni * 4 + r]);
          }
        if (m < p.M) {
          stvec<T, 8>(yg + (int64_t)m * p.ldy + n0, o0);
          stvec<T, 8>(yg + (int64_t)m * p.ldy + n0 + 8, o1);
        }
      }
    }
  }
}

template <typename T, int EPI, bool FP8>
int launch_gemm_tiled(const GemmTiledParams& p, hipStream_t st) {
  auto kern = gemm_tiled_kernel<T, EPI, FP8>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS);
  const int groups = (p.nsuper + 7) / 8;   // every XCD label gets the same number of 8 x 4 blocks
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * 8 * 32)), dim3(512), GT_LDS, st, p);
  MI355_CHECK_LAUNCH(EPI == 2 ? "mi355_qkv_proj_rope_tiled" : FP8 ? "mi355_fp8_scaled_mm_tiled" : "mi355_linear_tiled");
  return MI355_OK;
}

void fill_tiles(GemmTiledParams& p, int64_t m, int64_t n, int epilogue) {
  p.positions = nullptr; p.cos_sin = nullptr; p.cs_f32 = 0; p.hq = p.hkv = 0;
  p.k_buffer = p.v_buffer = nullptr; p.loc = nullptr; p.loc_i32 = 0; p.kv_slot = p.kv_head = 0;
  const int ncols = epilogue == 1 ? (int)(n / 2) : (int)n;
  p.ldy = ncols;
  p.mt = cdiv(m, 256);
  p.nt = epilogue == 1 ? cdiv(ncols, 128) : cdiv(ncols, 256);
  p.snt = cdiv(p.nt, 4);
  p.nsuper = cdiv(p.mt, 8) * p.snt;
}

}  // namespace
}  // namespace mi355

using namespace mi355;

extern "C" int mi355_linear_tiled_supported(int64_t m, int64_t n, int64_t k, int dtype, int epilogue) {
  if (dtype != MI355_BF16 && dtype != MI355_F16) return 0;
  if (m < 1 || n < 16 || k < 128 || (k & 63) != 0 || m > (1 << 24) || n > (1 << 24)) return 0;
  return ((epilogue == 0 || epilogue == 1) && n % 16 == 0) ? 1 : 0;   // silu-mul: 8 gate + 8 up columns per lane
}

extern "C" int mi355_linear_tiled(void* y, const void* x, const void* w, const void* bias, int64_t m,
                                  int64_t n, int64_t k, int64_t x_stride, int dtype, int epilogue,
                                  void* stream) {
  MI355_REQUIRE(y && x && w, MI355_ERR_INVALID, "mi355_linear_tiled: null pointer");
  MI355_REQUIRE(mi355_linear_tiled_supported(m, n, k, dtype, epilogue), MI355_ERR_UNSUPPORTED,
                "mi355_linear_tiled: needs a 16-bit dtype, K %% 64 == 0, K >= 128, N %% 16 == 0 (m=%ld n=%ld k=%ld "
                "dtype=%d epilogue=%d)", (long)m, (long)n, (long)k, dtype, epilogue);
  MI355_REQUIRE(x_stride >= k && x_stride % 8 == 0, MI355_ERR_INVALID,
                "mi355_linear_tiled: x_stride must be >= K and a multiple of 8 elements");
  MI355_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && (bias == nullptr || aligned16(bias)),
                MI355_ERR_INVALID, "mi355_linear_tiled: pointers must be 16-byte aligned");
  MI355_REQUIRE(epilogue == 0 || bias == nullptr, MI355_ERR_UNSUPPORTED,
                "mi355_linear_tiled: no bias with the SiLU-and-mul epilogue");
  GemmTiledParams p;
  p.x = x; p.w = w; p.bias = bias; p.y = y;
  p.x_scale = nullptr; p.w_scale = nullptr;
  p.M = (int)m; p.N = (int)n; p.K = (int)k;
  p.ldx = x_stride;
  fill_tiles(p, m, n, epilogue);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI355_BF16)
    return epilogue == 1 ? launch_gemm_tiled<bf16_t, 1, false>(p, st) : launch_gemm_tiled<bf16_t, 0, false>(p, st);
  return epilogue == 1 ? launch_gemm_tiled<f16_t, 1, false>(p, st) : launch_gemm_tiled<f16_t, 0, false>(p, st);
}

extern "C" int mi355_fp8_scaled_mm_tiled_supported(int64_t m, int64_t n, int64_t k, int out_dtype, int epilogue) {
  if (out_dtype != MI355_BF16 && out_dtype != MI355_F16) return 0;
  if (m < 1 || n < 16 || k < 256 || (k & 127) != 0 || m > (1 << 24) || n > (1 << 24)) return 0;
  return ((epilogue == 0 || epilogue == 1) && n % 16 == 0) ? 1 : 0;
}

extern "C" int mi355_fp8_scaled_mm_tiled(void* y, const uint8_t* xq, const uint8_t* wq, const float* x_scale,
                                         const float* w_scale, const void* bias, int64_t m, int64_t n,
                                         int64_t k, int out_dtype, int epilogue, void* stream) {
  MI355_REQUIRE(y && xq && wq && x_scale && w_scale, MI355_ERR_INVALID, "mi355_fp8_scaled_mm_tiled: null pointer");
  MI355_REQUIRE(mi355_fp8_scaled_mm_tiled_supported(m, n, k, out_dtype, epilogue), MI355_ERR_UNSUPPORTED,
                "mi355_fp8_scaled_mm_tiled: needs a 16-bit output dtype, K %% 128 == 0, K >= 256, N %% 16 == 0 "
                "(m=%ld n=%ld k=%ld out_dtype=%d epilogue=%d)", (long)m, (long)n, (long)k, out_dtype, epilogue);
  MI355_REQUIRE(aligned16(xq) && aligned16(wq) && aligned16(y) && (bias == nullptr || aligned16(bias)),
                MI355_ERR_INVALID, "mi355_fp8_scaled_mm_tiled: pointers must be 16-byte aligned");
  MI355_REQUIRE(epilogue == 0 || bias == nullptr, MI355_ERR_UNSUPPORTED,
                "mi355_fp8_scaled_mm_tiled: no bias with the SiLU-and-mul epilogue");
  GemmTiledParams p;
  p.x = xq; p.w = wq; p.bias = bias; p.y = y;
  p.x_scale = x_scale; p.w_scale = w_scale;
  p.M = (int)m; p.N = (int)n; p.K = (int)k;
  p.ldx = k;
  fill_tiles(p, m, n, epilogue);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == MI355_BF16)
    return epilogue == 1 ? launch_gemm_tiled<bf16_t, 1, true>(p, st) : launch_gemm_tiled<bf16_t, 0, true>(p, st);
  return epilogue == 1 ? launch_gemm_tiled<f16_t, 1, true>(p, st) : launch_gemm_tiled<f16_t, 0, true>(p, st);
}

extern "C" int mi355_qkv_proj_rope_tiled_supported(int64_t m, int64_t k, int hq, int hkv, int head_dim,
                                                   int rotary_dim, int is_neox, int dtype, int fp8) {
  if (dtype != MI355_BF16 && dtype != MI355_F16) return 0;
  if (head_dim != 128 || rotary_dim != 128 || !is_neox || hq < 1 || hkv < 1 || ((hq + 2 * hkv) & 1)) return 0;
  if (m < 1 || m > (1 << 24)) return 0;
  return fp8 ? (k >= 256 && (k & 127) == 0) : (k >= 128 && (k & 63) == 0);
}

extern "C" int mi355_qkv_proj_rope_tiled(void* qkv, const void* x, const void* w, const float* x_scale,
                                         const float* w_scale, int64_t m, int64_t k, int64_t x_stride,
                                         int hq, int hkv, int head_dim, const int64_t* positions,
                                         const void* cos_sin_cache, int cache_dtype, int rotary_dim, int is_neox,
                                         void* k_buffer, void* v_buffer, const void* loc, int loc_itype,
                                         int64_t num_slots, int dtype, void* stream) {
  const bool fp8 = x_scale != nullptr;
  const int kv_flags = dtype;
  dtype &= ~(MI355_KV_HEAD_MAJOR | MI355_KV_INTERLEAVED);
  MI355_REQUIRE(qkv && x && w && positions && cos_sin_cache && k_buffer && v_buffer && loc, MI355_ERR_INVALID,
                "mi355_qkv_proj_rope_tiled: null pointer");
  MI355_REQUIRE(!fp8 || w_scale, MI355_ERR_INVALID, "mi355_qkv_proj_rope_tiled: x_scale without w_scale");
  MI355_REQUIRE(mi355_qkv_proj_rope_tiled_supported(m, k, hq, hkv, head_dim, rotary_dim, is_neox, dtype, fp8),
                MI355_ERR_UNSUPPORTED,
                "mi355_qkv_proj_rope_tiled: needs a 16-bit dtype, head_dim = rotary_dim = 128, neox style, an even "
                "number of heads, K %% 64 == 0 (fp8: %% 128) (m=%ld k=%ld hq=%d hkv=%d head_dim=%d rotary_dim=%d)",
                (long)m, (long)k, hq, hkv, head_dim, rotary_dim);
  MI355_REQUIRE(cache_dtype == MI355_F32 || cache_dtype == dtype, MI355_ERR_UNSUPPORTED,
                "mi355_qkv_proj_rope_tiled: cos_sin_cache must be fp32 or the tensor dtype");
  MI355_REQUIRE(loc_itype == MI355_I32 || loc_itype == MI355_I64, MI355_ERR_INVALID,
                "mi355_qkv_proj_rope_tiled: loc must be int32 or int64");
  MI355_REQUIRE(fp8 ? x_stride == k : (x_stride >= k && x_stride % 8 == 0), MI355_ERR_INVALID,
                "mi355_qkv_proj_rope_tiled: bad x_stride");
  MI355_REQUIRE(aligned16(x) && aligned16(w) && aligned16(qkv) && aligned16(cos_sin_cache) && aligned16(k_buffer) &&
                    aligned16(v_buffer) && num_slots > 0 && num_slots < (1ll << 31),
                MI355_ERR_INVALID, "mi355_qkv_proj_rope_tiled: pointers must be 16-byte aligned");
  const int64_t n = (int64_t)(hq + 2 * hkv) * 128;
  GemmTiledParams p;
  p.x = x; p.w = w; p.bias = nullptr; p.y = qkv;
  p.x_scale = x_scale; p.w_scale = w_scale;
  p.M = (int)m; p.N = (int)n; p.K = (int)k;
  p.ldx = x_stride;
  fill_tiles(p, m, n, 0);
  p.positions = positions; p.cos_sin = cos_sin_cache; p.cs_f32 = cache_dtype == MI355_F32;
  p.hq = hq; p.hkv = hkv; p.k_buffer = k_buffer; p.v_buffer = v_buffer; p.loc = loc;
  kv_strides(kv_flags, num_slots, hkv, 128, p.kv_slot, p.kv_head);
  p.loc_i32 = loc_itype == MI355_I32;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI355_BF16)
    return fp8 ? launch_gemm_tiled<bf16_t, 2, true>(p, st) : launch_gemm_tiled<bf16_t, 2, false>(p, st);
  return fp8 ? launch_gemm_tiled<f16_t, 2, true>(p, st) : launch_gemm_tiled<f16_t, 2, false>(p, st);
}
