// Extend / prefill attention: new tokens attend to (1) their sequence's cached prefix, gathered
// through kv_indices from the paged KV cache, and (2) the new tokens themselves (causal
// triangle), with one online softmax across both phases.
//
// Structure: a workgroup (4 waves) owns a tile of 64 "rows" = TM query tokens x Gc query heads
// that share one kv head, so every K/V tile staged in LDS is reused by all heads of the group.
// K/V tiles of BN keys are gathered with 16-byte loads into padded LDS rows (bank-conflict free
// for the access patterns below).  Scores and P.V are register-blocked fp32 VALU FMAs (each lane
// a 4 x KJ score block, then a 4 x DPL output block), which keeps exact-fp32 accumulation for
// every dtype -- this is also the fp32 parity path.  Row max uses xor-shuffles across the 16
// lanes that share a row; the row sum is reduced once at the end.
#include "common.cuh"
#include <type_traits>
#include <math.h>

namespace mi355 {

static constexpr float kLog2eE = 1.4426950408889634f;

struct ExtendParams {
  const int32_t* qo_indptr;
  const int32_t* kv_indptr;
  const int32_t* kv_indices;
  const uint8_t* custom_mask;
  const int64_t* mask_indptr;
  int64_t q_stride, k_stride, v_stride;
  int is_causal, skip_prefix_custom_mask, hq, hkv, D, Dv, sliding_window;
  float sm_scale, logit_cap;
  // the cache buffers: elements between consecutive slots / consecutive kv heads ([slots, hkv, d] or head-major)
  int64_t kb_slot, kb_head, vb_slot, vb_head;
  int kv8;   // the cache holds e4m3fn bytes behind the element pointers (strides count bytes); read by the VALU kernel
};

// DP: padded head dim (>= D, Dv; multiple of 64). BN: keys per tile. VEC: 16-B global loads.
template <typename T, int DP, int BN, bool VEC>
__global__ __launch_bounds__(256, 2) void extend_attn_kernel(const T* __restrict__ q_ext,
                                                          const T* __restrict__ k_ext,
                                                          const T* __restrict__ v_ext,
                                                          T* __restrict__ o_ext,
                                                          const T* __restrict__ kbuf,
                                                          const T* __restrict__ vbuf,
                                                          ExtendParams p) {
  constexpr int VE = V16<T>::n;
  constexpr int LROW = DP + VE;          // padded LDS row (elements): +16 B
  constexpr int KJ = BN / 16;            // keys per lane in the score block
  constexpr int DPL = DP / 16;           // output features per lane
  constexpr int PROW = BN + 4;           // padded P row (floats)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* qs = reinterpret_cast<T*>(smem_raw);            // [64][LROW]
  T* ks = qs + 64 * LROW;                            // [BN][LROW]
  T* vs = ks + BN * LROW;                            // [BN][LROW]
  float* ps = reinterpret_cast<float*>(vs + BN * LROW);  // [64][PROW]

  const int gtotal = p.hq / p.hkv;
  const int Gc = gtotal < 64 ? gtotal : 64;
  const int chunks_per_kv = gtotal / Gc;
  const int TM = 64 / Gc;
  const int kvh = blockIdx.y / chunks_per_kv;
  const int head0 = kvh * gtotal + (blockIdx.y % chunks_per_kv) * Gc;
  const int b = blockIdx.z;
  const int q_start = p.qo_indptr[b];
  const int ext_len = p.qo_indptr[b + 1] - q_start;
  const int t0 = blockIdx.x * TM;
  if (t0 >= ext_len) return;
  const int kv_start = p.kv_indptr[b];
  const int prefix_len = p.kv_indptr[b + 1] - kv_start;
  const int seq_len = prefix_len + ext_len;
  const int D = p.D, Dv = p.Dv;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, rg = lane >> 4;  // c16: key / feature column group; rg: row group

  // rows of this lane: r_i = 16*wave + rg + 4*i
  int row_tok[4], row_head[4];
  bool row_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 16 * wave + rg + 4 * i;
    row_tok[i] = r / Gc;
    row_head[i] = r - row_tok[i] * Gc;
    row_ok[i] = row_tok[i] < TM && t0 + row_tok[i] < ext_len;
  }

  // ---- stage Q (zero padded to DP, rows beyond the tile zero) ----
  for (int pidx = threadIdx.x; pidx < 64 * (DP / VE); pidx += 256) {
    const int r = pidx / (DP / VE), c = (pidx % (DP / VE)) * VE;
    const int tq = r / Gc, g = r - tq * Gc;
    Vec<T, VE> val;
#pragma unroll
    for (int j = 0; j < VE; ++j) val.v[j] = (T)0.f;
    if (tq < TM && t0 + tq < ext_len) {
      const T* src = q_ext + (int64_t)(q_start + t0 + tq) * p.q_stride + (int64_t)(head0 + g) * D;
      if (VEC && c + VE <= D) {
        val = ldvec<T, VE>(src + c);
      } else {
#pragma unroll
        for (int j = 0; j < VE; ++j)
          if (c + j < D) val.v[j] = src[c + j];
      }
    }
    stvec<T, VE>(qs + r * LROW + c, val);
  }

  float m[4], l[4], acc[4][DPL];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m[i] = -INFINITY;
    l[i] = 0.f;
#pragma unroll
    for (int d = 0; d < DPL; ++d) acc[i][d] = 0.f;
  }

  const bool cap = p.logit_cap > 0.f;
  // keys of phase 2 this tile needs: causal -> up to its last token
  const int tile_last_tok = (t0 + TM < ext_len ? t0 + TM : ext_len) - 1;
  const int ext_keys = p.is_causal ? tile_last_tok + 1 : ext_len;

  for (int phase = 0; phase < 2; ++phase) {
    const int nkeys = phase == 0 ? prefix_len : ext_keys;
    for (int kb = 0; kb < nkeys; kb += BN) {
      __syncthreads();  // previous tile fully consumed (also covers the Q staging)
      // ---- stage K and V tiles ----
      for (int pidx = threadIdx.x; pidx < BN * (DP / VE); pidx += 256) {
        const int kr = pidx / (DP / VE), c = (pidx % (DP / VE)) * VE;
        const int key = kb + kr;
        Vec<T, VE> kval, vval;
#pragma unroll
        for (int j = 0; j < VE; ++j) {
          kval.v[j] = (T)0.f;
          vval.v[j] = (T)0.f;
        }
        if (key < nkeys) {
          const T *ksrc, *vsrc;
          if (phase == 0) {
            const int64_t slot = p.kv_indices[kv_start + key];
            ksrc = kbuf + slot * p.kb_slot + kvh * p.kb_head;
            vsrc = vbuf + slot * p.vb_slot + kvh * p.vb_head;
          } else {
            ksrc = k_ext + (int64_t)(q_start + key) * p.k_stride + (int64_t)kvh * D;
            vsrc = v_ext + (int64_t)(q_start + key) * p.v_stride + (int64_t)kvh * Dv;
          }
          if (phase == 0 && p.kv8) {   // e4m3fn prefix rows, widened exactly
            const int64_t slot = p.kv_indices[kv_start + key];
            const uint8_t* k8 = reinterpret_cast<const uint8_t*>(kbuf) + slot * p.kb_slot + kvh * p.kb_head;
            const uint8_t* v8 = reinterpret_cast<const uint8_t*>(vbuf) + slot * p.vb_slot + kvh * p.vb_head;
#pragma unroll
            for (int j = 0; j < VE; ++j) {
              if (c + j < D) kval.v[j] = from_f32<T>(__builtin_amdgcn_cvt_f32_fp8((int)k8[c + j], 0));
              if (c + j < Dv) vval.v[j] = from_f32<T>(__builtin_amdgcn_cvt_f32_fp8((int)v8[c + j], 0));
            }
          } else {
          if (VEC && c + VE <= D) {
            kval = ldvec<T, VE>(ksrc + c);
          } else {
#pragma unroll
            for (int j = 0; j < VE; ++j)
              if (c + j < D) kval.v[j] = ksrc[c + j];
          }
          if (VEC && c + VE <= Dv) {
            vval = ldvec<T, VE>(vsrc + c);
          } else {
#pragma unroll
            for (int j = 0; j < VE; ++j)
              if (c + j < Dv) vval.v[j] = vsrc[c + j];
          }
          }
        }
        stvec<T, VE>(ks + kr * LROW + c, kval);
        stvec<T, VE>(vs + kr * LROW + c, vval);
      }
      __syncthreads();

      // ---- scores: s[i][j] = q[row_i] . k[key c16 + 16 j] ----
      float s[4][KJ];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < KJ; ++j) s[i][j] = 0.f;
      const T* qrow = qs + (16 * wave + rg) * LROW;
      const T* krow = ks + c16 * LROW;
#pragma unroll 1
      for (int c = 0; c < DP; c += VE) {
        float qf[4][VE], kf[KJ][VE];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          Vec<T, VE> t = ldvec<T, VE>(qrow + 4 * i * LROW + c);
#pragma unroll
          for (int e = 0; e < VE; ++e) qf[i][e] = to_f32(t.v[e]);
        }
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
          Vec<T, VE> t = ldvec<T, VE>(krow + 16 * j * LROW + c);
#pragma unroll
          for (int e = 0; e < VE; ++e) kf[j][e] = to_f32(t.v[e]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int e = 0; e < VE; ++e) s[i][j] = fmaf(qf[i][e], kf[j][e], s[i][j]);
      }

      // ---- scale, cap, mask; online softmax ----
      float alpha[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tq = t0 + row_tok[i];  // extend-local query index
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
          const int key = kb + c16 + 16 * j;
          float x = s[i][j] * p.sm_scale;
          if (cap) x = p.logit_cap * tanhf(x / p.logit_cap);
          x *= kLog2eE;
          bool keep = row_ok[i] && key < nkeys;
          if (phase == 0) {
            if (p.custom_mask && !p.skip_prefix_custom_mask && keep)
              keep = p.custom_mask[p.mask_indptr[b] + (int64_t)tq * seq_len + key] != 0;
            if (p.sliding_window > 0) keep = keep && tq <= key + p.sliding_window;
          } else {
            if (p.custom_mask) {
              if (keep)
                keep = p.custom_mask[p.mask_indptr[b] + (int64_t)tq * seq_len + prefix_len + key] != 0;
            } else if (p.is_causal) {
              keep = keep && key <= tq;
            }
          }
          x = keep ? x : -INFINITY;
          s[i][j] = x;
          mx = fmaxf(mx, x);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        const float mnew = fmaxf(m[i], mx);
        const float ms = mnew == -INFINITY ? 0.f : mnew;
        alpha[i] = exp2f(m[i] - ms);
        m[i] = mnew;
        float psum = 0.f;
        float* prow = ps + (16 * wave + rg + 4 * i) * PROW;
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
          const float pe = exp2f(s[i][j] - ms);
          psum += pe;
          prow[c16 + 16 * j] = pe;
        }
        l[i] = l[i] * alpha[i] + psum;
      }
      __syncthreads();  // P visible (only same-wave rows are read, barrier keeps it simple)

      // ---- acc[i][d] = acc*alpha + sum_key P[row_i][key] * V[key][c16*DPL + d] ----
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < DPL; ++d) acc[i][d] *= alpha[i];
      const float* pbase = ps + (16 * wave + rg) * PROW;
      const T* vcol = vs + c16 * DPL;
#pragma unroll 1
      for (int k4 = 0; k4 < BN; k4 += 4) {
        float pf[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 t = *reinterpret_cast<const float4*>(pbase + 4 * i * PROW + k4);
          pf[i][0] = t.x; pf[i][1] = t.y; pf[i][2] = t.z; pf[i][3] = t.w;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          float vf[DPL];
          if constexpr (DPL >= VE) {
#pragma unroll
            for (int d0 = 0; d0 < DPL; d0 += VE) {
              Vec<T, VE> t = ldvec<T, VE>(vcol + (k4 + kk) * LROW + d0);
#pragma unroll
              for (int e = 0; e < VE; ++e) vf[d0 + e] = to_f32(t.v[e]);
            }
          } else {
            Vec<T, DPL> t = ldvec<T, DPL>(vcol + (k4 + kk) * LROW);
#pragma unroll
            for (int e = 0; e < DPL; ++e) vf[e] = to_f32(t.v[e]);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int d = 0; d < DPL; ++d) acc[i][d] = fmaf(pf[i][kk], vf[d], acc[i][d]);
        }
      }
    }
  }

  // ---- finish: reduce row sums over the 16 lanes of a row, normalise, store ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lt = l[i];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) lt += __shfl_xor(lt, off, 64);
    if (!row_ok[i]) continue;
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    T* dst = o_ext + ((int64_t)(q_start + t0 + row_tok[i]) * p.hq + head0 + row_head[i]) * Dv +
             c16 * DPL;
    if (VEC && c16 * DPL + DPL <= Dv) {
      Vec<T, DPL> out;
#pragma unroll
      for (int d = 0; d < DPL; ++d) out.v[d] = from_f32<T>(acc[i][d] * inv);
      stvec<T, DPL>(dst, out);
    } else {
#pragma unroll
      for (int d = 0; d < DPL; ++d)
        if (c16 * DPL + d < Dv) dst[d] = from_f32<T>(acc[i][d] * inv);
    }
  }
}

// ------------------------------------------------------------------------------------------
// MFMA flash-attention tile for the common prefill case: head_dim = v_head_dim = 128, 16-bit,
// causal or full, no custom mask / sliding window / logit cap.
//
// grid (q blocks of 64 rows, longest first; Hq; B), block 256 = 4 waves x 16 query rows.
// Per 64-key tile (K and V rows staged once in LDS for the 4 waves, next tile prefetched into
// registers while this one is computed):
//   S^T = K . Q^T   : A = K rows from LDS (ds_read_b128), B = Q^T kept in registers.  With the keys
//                     on the MFMA row axis, lane (query j = l & 15, c = l >> 4) owns the scores of
//                     ITS query for keys 16t + 4c + r: the softmax is lane-local apart from two
//                     cross-row maxima, and the probabilities are ALREADY the B operand of the
//                     next product (k index (c, e) <-> key 16 (e >> 2) + 4c + (e & 3)).
//   O^T += V^T . P^T: A = V^T fetched from the row-major V tile with ds_read_b64_tr_b16 (hardware
//                     transpose: 4 key rows x 16 features per 16 lanes), in the same key order.
// The output accumulator holds O^T[d = 16 dt + 4c + r][query j] and is rescaled per lane.
// ------------------------------------------------------------------------------------------
constexpr int XK_ROW = 128 + 16;   // K tile row (elements): 288 B.  Measured: SQ_LDS_BANK_CONFLICT is zero with 288-B rows and
                                   // 24 % of the LDS-active cycles with 272-B rows for the ds_read_b128 fragment reads
constexpr int XV_ROW = 128 + 16;   // V tile row: 288 B -> conflict-free transposed reads

// Three waves per SIMD (168 registers): the phases of a tile (LDS reads, MFMAs, softmax arithmetic,
// transposed reads, MFMAs) run back to back within a wave, so it is the number of resident waves that
// overlaps them: 2.17 -> 1.75 ms per layer against two waves.  To fit, the K fragments of a k-step
// are read right before their MFMAs (no second fragment buffer: the other waves cover that latency).
// KV8: the prefix lives in an e4m3fn cache (kbuf / vbuf are byte buffers); its rows are widened to T
// while they are staged.  The extend part (k_ext / v_ext) is always T.
// v_exp_f32 as is: the library exp2f wraps it in range scaling so that results below 2^-126 come out
// as denormals instead of zero (five instructions per call, and this kernel is bound by instruction
// issue).  A softmax weight that small contributes nothing at 16-bit output precision.
__device__ __forceinline__ float raw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <typename T, bool KV8 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void extend_attn_mfma_kernel(const T* __restrict__ q_ext,
                                                               const T* __restrict__ k_ext,
                                                               const T* __restrict__ v_ext,
                                                               T* __restrict__ o_ext,
                                                               const T* __restrict__ kbuf,
                                                               const T* __restrict__ vbuf,
                                                               ExtendParams p) {
  constexpr int D = 128;
  using frag = typename Mfma<T>::frag;
  __shared__ __attribute__((aligned(16))) T ks[64 * XK_ROW];
  __shared__ __attribute__((aligned(16))) T vs[64 * XV_ROW];

  const int qb = gridDim.x - 1 - blockIdx.x;   // long (late) blocks first
  const int h = blockIdx.y, b = blockIdx.z;
  const int kvh = h / (p.hq / p.hkv);
  const int q_start = p.qo_indptr[b];
  const int ext = p.qo_indptr[b + 1] - q_start;
  const int kv_start = p.kv_indptr[b];
  const int prefix = p.kv_indptr[b + 1] - kv_start;
  const int r0 = qb * 64;
  if (r0 >= ext) return;   // whole workgroup: EXEC stays full for the transposed reads below

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j16 = lane & 15, c4 = lane >> 4;
  const int qrow = r0 + wave * 16 + j16;                  // this lane's query (row of the extend part)
  const int qrow_c = qrow < ext ? qrow : ext - 1;
  const float qmul = p.sm_scale * kLog2eE;

  // B operand of S^T: Q[qrow][32 s + 8 c + 0..7]
  frag qf[4];
  {
    const T* qp = q_ext + (int64_t)(q_start + qrow_c) * p.q_stride + (int64_t)h * D + 8 * c4;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const frag*>(qp + 32 * s);
  }

  // keys this workgroup needs: [0, total) in the logical order prefix | extend
  const int last_row = min(r0 + 63, ext - 1);
  const int total = p.is_causal ? prefix + last_row + 1 : prefix + ext;
  const int n_tiles = (total + 63) / 64;
  const int my_limit = p.is_causal ? prefix + qrow_c : prefix + ext - 1;   // last visible key of this lane's query
  const int wave_min_limit = p.is_causal ? prefix + min(r0 + wave * 16, ext - 1) : prefix + ext - 1;

  // tile staging: thread -> rows (threadIdx.x >> 4) + 16 i, 16-B chunk (threadIdx.x & 15)
  const int srow = threadIdx.x >> 4, sch = threadIdx.x & 15;
  Vec<T, 8> kreg[4], vreg[4];
  // rows of the extend part are a fixed stride apart: one base pointer per staged row, no index math
  const T* k_lin = k_ext + (int64_t)(q_start - prefix + srow) * p.k_stride + (int64_t)kvh * D + sch * 8;
  const T* v_lin = v_ext + (int64_t)(q_start - prefix + srow) * p.v_stride + (int64_t)kvh * D + sch * 8;
  auto load_tile = [&](int kt) {
    if (kt * 64 >= prefix && kt * 64 + 63 < total) {   // whole tile in the extend part (workgroup-uniform)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t key = kt * 64 + 16 * i;
        kreg[i] = ldvec<T, 8>(k_lin + key * p.k_stride);
        vreg[i] = ldvec<T, 8>(v_lin + key * p.v_stride);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = kt * 64 + srow + 16 * i;
      key = key < total ? key : total - 1;
      if (KV8 && key < prefix) {
        const int64_t slot = p.kv_indices[kv_start + key];
        const uint8_t* kp8 = reinterpret_cast<const uint8_t*>(kbuf) + slot * p.kb_slot + kvh * p.kb_head + sch * 8;
        const uint8_t* vp8 = reinterpret_cast<const uint8_t*>(vbuf) + slot * p.vb_slot + kvh * p.vb_head + sch * 8;
        kreg[i] = cvt8_from_fp8<T>(*reinterpret_cast<const uint2*>(kp8));
        vreg[i] = cvt8_from_fp8<T>(*reinterpret_cast<const uint2*>(vp8));
        continue;
      }
      const T *kp, *vp;
      if (key < prefix) {
        const int64_t slot = p.kv_indices[kv_start + key];
        kp = kbuf + slot * p.kb_slot + kvh * p.kb_head;
        vp = vbuf + slot * p.vb_slot + kvh * p.vb_head;
      } else {
        const int64_t tok = q_start + key - prefix;
        kp = k_ext + tok * p.k_stride + (int64_t)kvh * D;
        vp = v_ext + tok * p.v_stride + (int64_t)kvh * D;
      }
      kreg[i] = ldvec<T, 8>(kp + sch * 8);
      vreg[i] = ldvec<T, 8>(vp + sch * 8);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      stvec<T, 8>(ks + (srow + 16 * i) * XK_ROW + sch * 8, kreg[i]);
      stvec<T, 8>(vs + (srow + 16 * i) * XV_ROW + sch * 8, vreg[i]);
    }
  };

  f32x4 ot[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;

  // LDS read bases
  const T* k_rd = ks + j16 * XK_ROW + 8 * c4;                          // + (16 t) rows, + 32 s
  // transposed read: lane 16 c + 4 q + pp supplies row (4 c + q), columns 4 pp .. 4 pp + 3
  const T* v_rd = vs + (4 * c4 + (j16 >> 2)) * XV_ROW + 4 * (j16 & 3);  // + (32 ks2 + 16 half) rows, + 16 dt

  load_tile(0);
  for (int kt = 0; kt < n_tiles; ++kt) {
    __syncthreads();            // previous tile fully consumed
    store_tile();
    __syncthreads();
    if (kt + 1 < n_tiles) load_tile(kt + 1);

    // S^T tile: st[t][r] = score of key 64 kt + 16 t + 4 c + r for this lane's query;
    // k-step major with the four key sub-tiles as independent accumulators
    f32x4 st[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      frag kf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) kf[t] = *reinterpret_cast<const frag*>(k_rd + (16 * t) * XK_ROW + 32 * s);
#pragma unroll
      for (int t = 0; t < 4; ++t) st[t] = Mfma<T>::run(kf[t], qf[s], st[t]);
    }
    const bool need_mask = kt * 64 + 63 > wave_min_limit;   // wave-uniform
    float mloc = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = st[t][r] * qmul;
        if (need_mask && kt * 64 + 16 * t + 4 * c4 + r > my_limit) x = -INFINITY;
        st[t][r] = x;
        mloc = fmaxf(mloc, x);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(m, mloc);
    const float ms = mnew == -INFINITY ? 0.f : mnew;
    const float alpha = raw_exp2(m - ms);
    m = mnew;
    float psum = 0.f;
    frag pb[2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pe = raw_exp2(st[t][r] - ms);
        // the denominator sums the fp32 weights; only the P.V operand is rounded to the value dtype
        // (as the reference's kernel: deno += sum(p), acc += dot(p.to(v.dtype), v),
        // triton_ops/extend_attention.py:190,200-201 and 271,281-282)
        psum += pe;
        pb[t >> 1][(t & 1) * 4 + r] = (T)pe;
      }
    lsum = lsum * alpha + psum;
    // multiplying by exactly 1 changes nothing: rescale only when some lane's maximum moved
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) ot[dt] *= alpha;
    }

#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const auto lo = TrRead<T>::run(v_rd + (32 * k2) * XV_ROW + 16 * dt);
        const auto hi = TrRead<T>::run(v_rd + (32 * k2 + 16) * XV_ROW + 16 * dt);
        frag va;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          va[e] = lo[e];
          va[4 + e] = hi[e];
        }
        ot[dt] = Mfma<T>::run(va, pb[k2], ot[dt]);
      }
  }

  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (qrow < ext) {
    const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
    T* op = o_ext + ((int64_t)(q_start + qrow) * p.hq + h) * D + 4 * c4;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      Vec<T, 4> w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w.v[r] = from_f32<T>(ot[dt][r] * inv);
      stvec<T, 4>(op + 16 * dt, w);
    }
  }
}

// ------------------------------------------------------------------------------------------
// The same prefill case on v_mfma_f32_32x32x16: a wave owns 32 query rows, a workgroup 128 rows of
// one query head (4 waves).  Why: with the 16x16x32 tile above every MFMA needs a fresh 1-KiB A
// fragment from LDS (K rows for the scores, V^T for the outputs), and twelve waves' fragment reads
// take as many LDS cycles as their MFMAs take matrix-core cycles; a 32x32x16 MFMA does twice the
// work on a fragment of the same size, and the loop issues half as many MFMA + LDS instructions.
//   S^T = K . Q^T    : A = 32 key rows (ds_read_b128), B = the wave's 32 queries, kept in registers
//                      (8 k-steps of 16 features).  Lane (query j = l & 31, h = l >> 5) then owns the
//                      scores of ITS query for keys 8 (i >> 2) + 4 h + (i & 3) of each 32-key half.
//   O^T += V^T . P^T : registers 8 u' .. 8 u' + 7 of a score tile, rounded to T, ARE the B fragment of
//                      the 16-key step u (element e <-> key 16 u + 8 (e >> 2) + 4 h + (e & 3)); the A
//                      fragment (V^T in that key order) comes from two transposed reads of the
//                      row-major V tile.
// K and V tiles are plain 256-byte rows with the chunk index XOR-ed by ((row & 3) << 2 | (row >> 2) & 3):
// conflict-free for the 32-row fragment reads and for the transposed reads.
// Workgroup -> work mapping: a 1-D grid; consecutive workgroups go round-robin over the 8 XCDs, so
// workgroup L serves the (sequence, kv head) pair number (L & 7) + 8 * ((L >> 3) / per_pair): every
// query block and every query head of a pair runs on ONE XCD and finds the pair's K/V in that XCD's
// L2 after the first touch (the 2-D grid fetched each K/V row from HBM 2.8 times).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int xor_image_off(int row, int ch) {   // bytes; ch = 16-byte chunk 0..15 of a 256-byte row
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// DQ: query / key head dimension, 128 or 192 (the multi-head form of MLA prefill: 128 nope + 64 rope features
// against 128-wide values).  Features 128 .. 191 of the K tile are a second 256-byte-row image (half of each row
// used) with the same chunk XOR, so the fragment reads of score steps 8 .. 11 are as conflict-free as the others.
template <typename T, bool KV8, int DQ>
__device__ __forceinline__ void extend_attn_mfma32_body(
    const T* __restrict__ q_ext, const T* __restrict__ k_ext, const T* __restrict__ v_ext, T* __restrict__ o_ext,
    const T* __restrict__ kbuf, const T* __restrict__ vbuf, const ExtendParams& p, int nqb, int npairs) {
  constexpr int D = 128;          // value head dimension = width of one tile image
  constexpr int NS = DQ / 16;     // score k-steps
  constexpr bool WIDE = DQ > 128;
  static_assert(DQ == 128 || (DQ == 192 && !KV8), "128, or 192 over a 16-bit cache");
  using frag = typename Mfma32<T>::frag;
  __shared__ __attribute__((aligned(16))) char ks[64 * 256];
  __shared__ __attribute__((aligned(16))) char ks2[WIDE ? 64 * 256 : 16];
  __shared__ __attribute__((aligned(16))) char vs[64 * 256];

  const int G = p.hq / p.hkv;
  const int per_pair = G * nqb;
  const int slot = blockIdx.x >> 3;
  const int pair = (blockIdx.x & 7) + 8 * (slot / per_pair);
  if (pair >= npairs) return;
  const int t_in_pair = slot % per_pair;
  const int qb = nqb - 1 - t_in_pair / G;          // long (late) blocks first
  const int b = pair / p.hkv, kvh = pair % p.hkv, h = kvh * G + t_in_pair % G;
  const int q_start = p.qo_indptr[b];
  const int ext = p.qo_indptr[b + 1] - q_start;
  const int kv_start = p.kv_indptr[b];
  const int prefix = p.kv_indptr[b + 1] - kv_start;
  const int r0 = qb * 128;
  if (r0 >= ext) return;   // whole workgroup: EXEC stays full for the transposed reads below

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j32 = lane & 31, hh = lane >> 5;
  const int qrow = r0 + wave * 32 + j32;                  // this lane's query (row of the extend part)
  const int qrow_c = qrow < ext ? qrow : ext - 1;
  const float qmul = p.sm_scale * kLog2eE;

  // B operand of S^T: Q[qrow][16 s + 8 h + 0..7]
  frag qf[NS];
  {
    const T* qp = q_ext + (int64_t)(q_start + qrow_c) * p.q_stride + (int64_t)h * DQ + 8 * hh;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const frag*>(qp + 16 * s);
  }

  const int last_row = min(r0 + 127, ext - 1);
  const int total = p.is_causal ? prefix + last_row + 1 : prefix + ext;
  const int n_tiles = (total + 63) / 64;
  const int my_limit = p.is_causal ? prefix + qrow_c : prefix + ext - 1;   // last visible key of this lane's query
  const int wave_min_limit = p.is_causal ? prefix + min(r0 + wave * 32, ext - 1) : prefix + ext - 1;
  const int wave_max_limit = p.is_causal ? prefix + min(r0 + wave * 32 + 31, ext - 1) : prefix + ext - 1;

  // tile staging: thread -> rows (threadIdx.x >> 4) + 16 i, 16-B chunk (threadIdx.x & 15)
  const int srow = threadIdx.x >> 4, sch = threadIdx.x & 15;
  Vec<T, 8> kreg[4], vreg[4], kreg2[WIDE ? 4 : 1];   // kreg2: features 128 + 8 sch .. + 7 (threads with sch < 8)
  const T* k_lin = k_ext + (int64_t)(q_start - prefix + srow) * p.k_stride + (int64_t)kvh * DQ + sch * 8;
  const T* v_lin = v_ext + (int64_t)(q_start - prefix + srow) * p.v_stride + (int64_t)kvh * D + sch * 8;
  auto load_tile = [&](int kt) {
    if (kt * 64 >= prefix && kt * 64 + 63 < total) {   // whole tile in the extend part (workgroup-uniform)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t key = kt * 64 + 16 * i;
        kreg[i] = ldvec<T, 8>(k_lin + key * p.k_stride);
        vreg[i] = ldvec<T, 8>(v_lin + key * p.v_stride);
        if constexpr (WIDE) {
          if (sch < 8) kreg2[i] = ldvec<T, 8>(k_lin + key * p.k_stride + 128);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = kt * 64 + srow + 16 * i;
      key = key < total ? key : total - 1;
      if (KV8 && key < prefix) {
        const int64_t slot_ = p.kv_indices[kv_start + key];
        const uint8_t* kp8 = reinterpret_cast<const uint8_t*>(kbuf) + slot_ * p.kb_slot + kvh * p.kb_head + sch * 8;
        const uint8_t* vp8 = reinterpret_cast<const uint8_t*>(vbuf) + slot_ * p.vb_slot + kvh * p.vb_head + sch * 8;
        kreg[i] = cvt8_from_fp8<T>(*reinterpret_cast<const uint2*>(kp8));
        vreg[i] = cvt8_from_fp8<T>(*reinterpret_cast<const uint2*>(vp8));
        continue;
      }
      const T *kp, *vp;
      if (key < prefix) {
        const int64_t slot_ = p.kv_indices[kv_start + key];
        kp = kbuf + slot_ * p.kb_slot + kvh * p.kb_head;
        vp = vbuf + slot_ * p.vb_slot + kvh * p.vb_head;
      } else {
        const int64_t tok = q_start + key - prefix;
        kp = k_ext + tok * p.k_stride + (int64_t)kvh * DQ;
        vp = v_ext + tok * p.v_stride + (int64_t)kvh * D;
      }
      kreg[i] = ldvec<T, 8>(kp + sch * 8);
      vreg[i] = ldvec<T, 8>(vp + sch * 8);
      if constexpr (WIDE) {
        if (sch < 8) kreg2[i] = ldvec<T, 8>(kp + 128 + sch * 8);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = xor_image_off(srow + 16 * i, sch);
      *reinterpret_cast<Vec<T, 8>*>(ks + off) = kreg[i];
      *reinterpret_cast<Vec<T, 8>*>(vs + off) = vreg[i];
      if constexpr (WIDE) {
        if (sch < 8) *reinterpret_cast<Vec<T, 8>*>(ks2 + off) = kreg2[i];
      }
    }
  };

  f32x16 ot[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) ot[dt][i] = 0.f;
  float m = -INFINITY, lsum = 0.f;

  // K fragment of (32-key half hf, k-step s): row 32 hf + j32, chunk 2 s + h.  The XOR term depends on
  // (row & 15) only, i.e. on the lane: chunk ^ fx = (2 s) ^ (h ^ fx).
  const int fx = ((j32 & 3) << 2) | ((j32 >> 2) & 3);
  const int kcx = hh ^ fx;
  const char* k_rd = ks + 256 * j32;                        // + 8192 hf + 16 * ((2 s) ^ kcx)
  const char* k_rd2 = ks2 + 256 * j32;                      // steps 8 .. : chunk 2 (s - 8) + h of the second image
  auto k_frag = [&](int s_, int hf) -> frag {
    const char* base = (WIDE && s_ >= 8) ? k_rd2 : k_rd;
    return *reinterpret_cast<const frag*>(base + 8192 * hf + 16 * ((2 * (s_ & 7)) ^ kcx));
  };
  // V^T fragment of (d tile dt, 16-key step u): lane group (l >> 4) = 2 h + dhalf serves features
  // 32 dt + 16 dhalf + 0..15; lane 4 q + pp of the group supplies row 16 u (+ 8) + 4 h + q, features 4 pp..4 pp + 3
  const int l16 = lane & 15, dhalf = (lane >> 4) & 1, tq = l16 >> 2, tpp = l16 & 3;
  const int vrow = 4 * hh + tq;                             // + 16 u (+ 8): (row & 3) = tq, (row >> 2) & 3 = (h + 2 hi) & 3
  const int vf_lo = (tq << 2) | (hh & 3), vf_hi = (tq << 2) | ((hh + 2) & 3);
  const int vch = 2 * dhalf + (tpp >> 1);                   // + 4 dt
  const char* v_rd = vs + 256 * vrow + 8 * (tpp & 1);

  load_tile(0);
  for (int kt = 0; kt < n_tiles; ++kt) {
    __syncthreads();            // previous tile fully consumed
    store_tile();
    __syncthreads();
    if (kt + 1 < n_tiles) load_tile(kt + 1);
    if (kt * 64 > wave_max_limit) continue;   // every key of the tile is masked for every query of this wave (wave-uniform)

    f32x16 st[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int i = 0; i < 16; ++i) st[hf][i] = 0.f;
    // LDS fragment reads run AHEAD of the MFMAs that use them (register rings, fenced): written
    // read-MFMA-read-MFMA, hipcc keeps that order and every MFMA waits a full LDS round trip
    // (>= 64 cycles against the MFMA's 32).
    frag kr[3][2];   // K fragments of k-steps s, s + 1, s + 2
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      kr[s][0] = k_frag(s, 0);
      kr[s][1] = k_frag(s, 1);
    }
    // the first V^T fragments too: the tile is already in LDS, and the softmax below covers their latency
    typename TrRead<T>::v4 vlo[4], vhi[4];   // ring of four (u, dt) fragments
    auto v_read = [&](int i, int slot_) {
      const int u = i >> 2, dt = i & 3;
      const int ch = 4 * dt + vch;
      vlo[slot_] = TrRead<T>::run(reinterpret_cast<const T*>(v_rd + 256 * (16 * u) + 16 * (ch ^ vf_lo)));
      vhi[slot_] = TrRead<T>::run(reinterpret_cast<const T*>(v_rd + 256 * (16 * u + 8) + 16 * (ch ^ vf_hi)));
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 2 < NS) {
        kr[(s + 2) % 3][0] = k_frag(s + 2, 0);
        kr[(s + 2) % 3][1] = k_frag(s + 2, 1);
      } else if (s == NS - 2) {
        v_read(0, 0);
        v_read(1, 1);
      } else {
        v_read(2, 2);
        v_read(3, 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      st[0] = Mfma32<T>::run(kr[s % 3][0], qf[s], st[0]);
      st[1] = Mfma32<T>::run(kr[s % 3][1], qf[s], st[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // Softmax on the raw scores: the scale (sm_scale * log2 e > 0, checked by the launcher) commutes with
    // the maximum and folds into the exponent's multiply-add.  Only tiles that cross the wave's diagonal
    // pay for masking (two instructions per score: the key offsets within a tile are compile-time
    // constants, the lane's limit is one subtraction per tile); the branch is wave-uniform.
    float mloc = -INFINITY;
    if (kt * 64 + 63 > wave_min_limit) {
      const int lim = my_limit - kt * 64 - 4 * hh;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float x = (32 * hf + 8 * (i >> 2) + (i & 3)) > lim ? -INFINITY : st[hf][i];
          st[hf][i] = x;
          mloc = fmaxf(mloc, x);
        }
    } else {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, st[hf][i]);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(m, mloc * qmul);
    const float ms = mnew == -INFINITY ? 0.f : mnew;
    const float alpha = raw_exp2(m - ms);
    m = mnew;
    float psum = 0.f;
    frag pb[4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pe = raw_exp2(__builtin_fmaf(st[hf][i], qmul, -ms));
        psum += pe;                          // fp32 weights in the denominator, T-rounded weights in P.V
        pb[2 * hf + (i >> 3)][i & 7] = (T)pe;
      }
    lsum = lsum * alpha + psum;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ot[dt] *= alpha;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {   // (u, dt) = (i >> 2, i & 3); fragment i + 4 is requested right behind MFMA i
      frag va;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        va[e] = vlo[i & 3][e];
        va[4 + e] = vhi[i & 3][e];
      }
      ot[i & 3] = Mfma32<T>::run(va, pb[i >> 2], ot[i & 3]);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 4 < 16) {
        v_read(i + 4, i & 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  lsum += __shfl_xor(lsum, 32, 64);
  if (qrow < ext) {
    const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
    T* op = o_ext + ((int64_t)(q_start + qrow) * p.hq + h) * D + 4 * hh;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        Vec<T, 4> w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w.v[r] = from_f32<T>(ot[dt][4 * rq + r] * inv);
        stvec<T, 4>(op + 32 * dt + 8 * rq, w);
      }
  }
}

// Two workgroups per CU (256 registers per lane) at head dimension 128; the 192-wide variant needs ~280 registers and
// runs one workgroup per CU instead of spilling (measured at 4 x 4096 tokens x 16 heads: 470 TFLOP/s against 412 with
// 24 spilled registers at two workgroups per CU; issuing the next tile's loads behind the score phase spills more).
template <typename T, bool KV8 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void extend_attn_mfma32_kernel(
    const T* __restrict__ q_ext, const T* __restrict__ k_ext, const T* __restrict__ v_ext, T* __restrict__ o_ext,
    const T* __restrict__ kbuf, const T* __restrict__ vbuf, ExtendParams p, int nqb, int npairs) {
  extend_attn_mfma32_body<T, KV8, 128>(q_ext, k_ext, v_ext, o_ext, kbuf, vbuf, p, nqb, npairs);
}
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void extend_attn_mfma32_wide_kernel(
    const T* __restrict__ q_ext, const T* __restrict__ k_ext, const T* __restrict__ v_ext, T* __restrict__ o_ext,
    const T* __restrict__ kbuf, const T* __restrict__ vbuf, ExtendParams p, int nqb, int npairs) {
  extend_attn_mfma32_body<T, false, 192>(q_ext, k_ext, v_ext, o_ext, kbuf, vbuf, p, nqb, npairs);
}

template <typename T, int DP, int BN>
static int launch_extend_cfg(const void* q, const void* k, const void* v, void* o, const void* kb,
                             const void* vb, const ExtendParams& p, int64_t bs,
                             int64_t max_len_extend, bool vec, hipStream_t st) {
  constexpr int VE = V16<T>::n;
  const int gtotal = p.hq / p.hkv;
  const int Gc = gtotal < 64 ? gtotal : 64;
  const int TM = 64 / Gc;
  const dim3 grid((unsigned)((max_len_extend + TM - 1) / TM), (unsigned)(p.hkv * (gtotal / Gc)),
                  (unsigned)bs);
  const size_t lds = (size_t)(64 + 2 * BN) * (DP + VE) * sizeof(T) + (size_t)64 * (BN + 4) * 4;
  if (vec) {
    auto kern = extend_attn_kernel<T, DP, BN, true>;
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const T*)q, (const T*)k, (const T*)v, (T*)o,
                       (const T*)kb, (const T*)vb, p);
  } else {
    auto kern = extend_attn_kernel<T, DP, BN, false>;
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const T*)q, (const T*)k, (const T*)v, (T*)o,
                       (const T*)kb, (const T*)vb, p);
  }
  return MI355_OK;
}

// 0 = 32x32x16 tile (product), 1 = the 16x16x32 tile above (ablation; development build, mi355_debug_set key 8)
#ifdef MI355_DEV_BUILD
int g_extend_variant = 0;
#else
static constexpr int g_extend_variant = 0;
#endif

template <typename T>
static int launch_extend(const void* q, const void* k, const void* v, void* o, const void* kb,
                         const void* vb, const ExtendParams& p, int64_t bs, int64_t max_len_extend,
                         bool kv8, hipStream_t st) {   // kv8: the KV cache holds e4m3fn bytes
  constexpr int VE = V16<T>::n;
  const int dmax = p.D > p.Dv ? p.D : p.Dv;
  const bool vec = p.D % VE == 0 && p.Dv % VE == 0 && p.q_stride % VE == 0 &&
                   p.k_stride % VE == 0 && p.v_stride % VE == 0 && aligned16(q) && aligned16(k) &&
                   aligned16(v) && aligned16(o) && aligned16(kb) && aligned16(vb);
  if constexpr (!std::is_same<T, float>::value) {
    if (vec && (p.D == 128 || (p.D == 192 && !kv8)) && p.Dv == 128 && !p.custom_mask && p.sliding_window <= 0 &&
        p.logit_cap <= 0.f && p.sm_scale > 0.f && g_extend_variant == 0) {   // the 32x32 kernel takes the maximum of unscaled scores
      // 128 query rows of one head per workgroup; (sequence, kv head) pairs dealt round-robin to the XCDs
      const int nqb = (int)((max_len_extend + 127) / 128);
      const int64_t npairs = bs * p.hkv;
      const int64_t wgs = 8 * ((npairs + 7) / 8) * (p.hq / p.hkv) * nqb;
      if (wgs < (int64_t)1 << 31) {
        const dim3 grid1((unsigned)wgs);
        if (p.D == 192)
          hipLaunchKernelGGL((extend_attn_mfma32_wide_kernel<T>), grid1, dim3(256), 0, st, (const T*)q, (const T*)k,
                             (const T*)v, (T*)o, (const T*)kb, (const T*)vb, p, nqb, (int)npairs);
        else if (kv8)
          hipLaunchKernelGGL((extend_attn_mfma32_kernel<T, true>), grid1, dim3(256), 0, st, (const T*)q, (const T*)k,
                             (const T*)v, (T*)o, (const T*)kb, (const T*)vb, p, nqb, (int)npairs);
        else
          hipLaunchKernelGGL((extend_attn_mfma32_kernel<T, false>), grid1, dim3(256), 0, st, (const T*)q, (const T*)k,
                             (const T*)v, (T*)o, (const T*)kb, (const T*)vb, p, nqb, (int)npairs);
        return MI355_OK;
      }
    }
    if (vec && p.D == 128 && p.Dv == 128 && !p.custom_mask && p.sliding_window <= 0 && p.logit_cap <= 0.f) {
      const dim3 grid((unsigned)((max_len_extend + 63) / 64), (unsigned)p.hq, (unsigned)bs);
      if (kv8)
        hipLaunchKernelGGL((extend_attn_mfma_kernel<T, true>), grid, dim3(256), 0, st, (const T*)q, (const T*)k,
                           (const T*)v, (T*)o, (const T*)kb, (const T*)vb, p);
      else
        hipLaunchKernelGGL((extend_attn_mfma_kernel<T, false>), grid, dim3(256), 0, st, (const T*)q, (const T*)k,
                           (const T*)v, (T*)o, (const T*)kb, (const T*)vb, p);
      return MI355_OK;
    }
  }
  if (dmax <= 64) return launch_extend_cfg<T, 64, 64>(q, k, v, o, kb, vb, p, bs, max_len_extend, vec, st);
  if (dmax <= 128) return launch_extend_cfg<T, 128, 64>(q, k, v, o, kb, vb, p, bs, max_len_extend, vec, st);
  if (dmax <= 256) {
    if (sizeof(T) == 4)
      return launch_extend_cfg<T, 256, 32>(q, k, v, o, kb, vb, p, bs, max_len_extend, vec, st);
    return launch_extend_cfg<T, 256, 64>(q, k, v, o, kb, vb, p, bs, max_len_extend, vec, st);
  }
  set_error("extend_attention: head_dim %d / v_head_dim %d > 256 not supported", p.D, p.Dv);
  return MI355_ERR_UNSUPPORTED;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_extend_attention(const void* q_extend, const void* k_extend, const void* v_extend,
                           void* o_extend, const void* k_buffer, const void* v_buffer,
                           const int32_t* qo_indptr, const int32_t* kv_indptr,
                           const int32_t* kv_indices, const uint8_t* custom_mask,
                           const int64_t* mask_indptr, int is_causal, int skip_prefix_custom_mask,
                           int64_t bs, int64_t max_len_extend, int hq, int hkv, int head_dim,
                           int v_head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                           int64_t num_slots, float sm_scale, float logit_cap,
                           int sliding_window_size, int dtype, void* stream) {
  MI355_REQUIRE(q_extend && k_extend && v_extend && o_extend && qo_indptr && kv_indptr,
                MI355_ERR_INVALID, "extend_attention: null pointer");
  MI355_REQUIRE(k_buffer && v_buffer && kv_indices, MI355_ERR_INVALID,
                "extend_attention: k_buffer/v_buffer/kv_indices must be valid pointers (may be "
                "unused when every prefix is empty, but must not be null)");
  MI355_REQUIRE(!custom_mask || mask_indptr, MI355_ERR_INVALID,
                "extend_attention: custom_mask needs mask_indptr");
  MI355_REQUIRE(bs >= 0 && max_len_extend >= 0 && hq > 0 && hkv > 0 && hq % hkv == 0 &&
                    head_dim > 0 && v_head_dim > 0,
                MI355_ERR_INVALID, "extend_attention: bad shape");
  const int gtotal = hq / hkv;
  MI355_REQUIRE(gtotal <= 64 ? true : gtotal % 64 == 0, MI355_ERR_UNSUPPORTED,
                "extend_attention: q heads per kv head (%d) must be <= 64 or a multiple of 64",
                gtotal);
  MI355_REQUIRE(q_stride >= (int64_t)hq * head_dim && k_stride >= (int64_t)hkv * head_dim &&
                    v_stride >= (int64_t)hkv * v_head_dim,
                MI355_ERR_INVALID, "extend_attention: row strides smaller than rows");
  MI355_REQUIRE(bs <= 65535, MI355_ERR_UNSUPPORTED, "extend_attention: batch > 65535");
  if (bs == 0 || max_len_extend == 0) return MI355_OK;
  ExtendParams p;
  p.qo_indptr = qo_indptr;
  p.kv_indptr = kv_indptr;
  p.kv_indices = kv_indices;
  p.custom_mask = custom_mask;
  p.mask_indptr = mask_indptr;
  p.q_stride = q_stride;
  p.k_stride = k_stride;
  p.v_stride = v_stride;
  p.is_causal = is_causal;
  p.skip_prefix_custom_mask = skip_prefix_custom_mask;
  p.hq = hq;
  p.hkv = hkv;
  p.D = head_dim;
  p.Dv = v_head_dim;
  p.sliding_window = sliding_window_size;
  p.sm_scale = sm_scale;
  p.logit_cap = logit_cap;
  MI355_REQUIRE(!(dtype & MI355_KV_INTERLEAVED) || ((dtype & MI355_KV_HEAD_MAJOR) && head_dim == v_head_dim), MI355_ERR_INVALID,
                "extend_attention: an interleaved cache is head-major with head_dim == v_head_dim");
  kv_strides(dtype, num_slots, hkv, head_dim, p.kb_slot, p.kb_head);
  kv_strides(dtype, num_slots, hkv, v_head_dim, p.vb_slot, p.vb_head);
  int rc = MI355_OK;
  const bool kv8 = (dtype & MI355_KV_FP8) != 0;
  p.kv8 = kv8 ? 1 : 0;
  dtype &= 0xff;
  MI355_DISPATCH_DTYPE(dtype, "extend_attention",
                       rc = launch_extend<T>(q_extend, k_extend, v_extend, o_extend, k_buffer,
                                             v_buffer, p, bs, max_len_extend, kv8, (hipStream_t)stream));
  if (rc != MI355_OK) return rc;
  MI355_CHECK_LAUNCH("extend_attention");
  return MI355_OK;
}

}  // extern "C"
