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
#include <cmath>
#include "common.cuh"
#include <math.h>
#include <type_traits>
#include <mutex>

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
  // The cache as the caller laid it out: elements between consecutive slots and between consecutive kv heads
  // ([slots, hkv, d]: hkv * d and d; head-major [hkv, slots, d]: d and slots * d).
  int64_t k_slot_stride, v_slot_stride;
  int64_t k_head_stride, v_head_stride;
  // Arrival counters of the in-kernel merge, one per (sequence, blockIdx.x), zero between launches; null:
  // the splits are merged by decode_merge_kernel in a second launch.  Only the kernels that say so use them.
  unsigned* tickets = nullptr;
  // Per-sequence split plan: when non-null, one int32 on the device = the batch's keys per split
  // (ceil(longest sequence / num_splits), written by the host before the launch).  Sequence b is then cut
  // into ceil(len(b) / that) splits instead of num_splits; its remaining split workgroups find nothing.
  const int32_t* chunk_keys = nullptr;
};

// Keys per split of a sequence of `len` keys: the even cut into num_splits, or the batch's chunk when
// that is larger (never smaller: num_splits chunks always cover the sequence).
__device__ __forceinline__ int64_t split_chunk(const DecodeParams& p, int64_t len) {
  int64_t chunk = (len + p.num_splits - 1) / p.num_splits;
  if (p.chunk_keys) {
    const int64_t c = *p.chunk_keys;
    chunk = c > chunk ? c : chunk;
  }
  return chunk;
}

// Inputs of the fused decode step: the qkv projection is still fp32 split-K partials, RoPE has
// not been applied and the new token's K/V rows are not in the cache yet.
struct FusedQkv {
  const float* part;          // [nsplit][B][row] partials of the qkv GEMM, row = (hq + 2 hkv) * 128
  int nsplit;
  int64_t split_stride, row;  // elements
  const float* xs;            // fp8 qkv GEMM: per-token activation scales [B] and per-channel weight
  const float* ws;            //   scales [row]; both null for a 16-bit GEMM
  const int64_t* positions;   // [B]
  const int64_t* out_loc;     // [B] cache slot of the new token (padded rows of a captured batch name the sink slot)
  const void* cos_sin;        // [max_pos][128]: cos (64) | sin (64), fp32 or the tensor dtype
  int cache_f32;
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

// ---- stage 2 of split-KV decoding: o[b, h, :] = sum_s w_s part_o[b, h, s, :] / sum_s w_s with
// w_s = exp(lse_s - max lse).  ONE statement of the arithmetic, shared by the merge launch and by the
// in-kernel merge below, so that both give the same bits.
__device__ __forceinline__ float merge_weights(const float* lse, int num_splits, float* w) {
  float mx = -INFINITY;
  for (int s = 0; s < num_splits; ++s) mx = fmaxf(mx, lse[s]);
  const float ms = safe_m(mx);
  float tot = 0.f;
  for (int s = 0; s < num_splits; ++s) {
    const float e = expf(lse[s] - ms);
    if (w) w[s] = e;
    tot += e;
  }
  return tot;
}
__device__ __forceinline__ float merge_step(float w, float part, float a) { return __builtin_fmaf(w, part, a); }
__device__ __forceinline__ float merge_finish(float a, float tot) { return tot > 0.f ? a / tot : 0.f; }

// Buffer resource over `bytes` at `ptr`; aux = 16 (sc1) on a store writes through to memory, on a load
// never returns a line that another XCD's store has made stale.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ auto wt_rsrc(const void* ptr, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}
constexpr int MERGE_MAX_SPLITS = 64;   // the in-kernel merge stages (heads of a workgroup) x (splits) weights in LDS

// In-kernel merge of the KV splits (split-KV decoding without the second launch).  Every workgroup of
// a (sequence, blockIdx.x) pair calls this after storing its partial outputs and log-sum-exp WRITE-THROUGH
// (sc1); the stores are drained, the workgroup takes a ticket, and the workgroup that draws the last
// ticket reads all splits back (sc1 loads), merges its `nheads` heads x Dv features with the arithmetic
// of decode_merge_kernel, writes o and puts the ticket back to zero for the next launch.
// ticket_idx: one counter per (sequence, head group of the workgroup).
// lds: >= 8 + nheads * (num_splits + 1) floats that no thread of the workgroup still reads.
template <typename T, int IT>
__device__ __forceinline__ void merge_splits_if_last(const DecodeParams& p, T* __restrict__ o, int b, int head0,
                                                     int nheads, int Dv, int ticket_idx, float* lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // every wave's partial stores have reached memory; `lds` is free
  int* flag = reinterpret_cast<int*>(lds);
  unsigned* ticket = p.tickets + ticket_idx;
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = old == (unsigned)(p.num_splits - 1);
  }
  __syncthreads();
  if (!*flag) return;
  const int S = p.num_splits;
  float* w = lds + 8;             // [nheads][S]
  float* tot = w + nheads * S;    // [nheads]
  const int64_t bh0 = (int64_t)b * p.hq + head0;
  const auto ro = wt_rsrc(p.part_o + bh0 * S * Dv, nheads * S * Dv * 4);
  const int per_head = Dv >> 2, n_items = nheads * per_head;
  // A thread merges IT four-feature items at a time, eight splits each: IT x 8 loads per memory round
  // trip.  The first round's loads do not depend on the weights and are issued ahead of them.
  int gi[IT], di[IT];
  f32x4 v[IT][8];   // (whole vectors are cast: a bit_cast of ONE element of the integer vector reads element 0 for all four)
  auto place = [&](int first) {
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      const int at = first + t * (int)blockDim.x;
      const int item = at < n_items ? at : 0;   // (a clamped item is loaded, never stored)
      gi[t] = item / per_head;
      di[t] = 4 * (item - gi[t] * per_head);
    }
  };
  // (up to four splits -- the plans of full batches -- go out as a round of four: the clamped loads of a wider
  // round are real requests to memory)
  const bool narrow = S <= 4;
  auto load_round = [&](int s0) {
#pragma unroll
    for (int t = 0; t < IT; ++t)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (narrow && u >= 4) continue;   // (wave-uniform)
        const int sp = s0 + u < S ? s0 + u : S - 1;
        v[t][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ro, ((gi[t] * S + sp) * Dv + di[t]) * 4, 0, 16));
      }
  };
  place(threadIdx.x);
  load_round(0);
  const auto rl = wt_rsrc(p.part_lse + bh0 * S, nheads * S * 4);
  for (int i = threadIdx.x; i < nheads * S; i += blockDim.x)
    w[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, i * 4, 0, 16));
  __syncthreads();
  for (int g = threadIdx.x; g < nheads; g += blockDim.x) tot[g] = merge_weights(w + g * S, S, w + g * S);   // in place
  __syncthreads();
  for (int first = threadIdx.x; first < n_items; first += IT * blockDim.x) {
    float a[IT][4];
#pragma unroll
    for (int t = 0; t < IT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[t][j] = 0.f;
    if (first != (int)threadIdx.x) {
      place(first);
      load_round(0);
    }
    for (int s0 = 0; s0 < S; s0 += 8) {
      if (s0 > 0) load_round(s0);
#pragma unroll
      for (int t = 0; t < IT; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (narrow && u >= 4) continue;   // (never loaded)
          if (s0 + u < S) {
            const float ws = w[gi[t] * S + s0 + u];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[t][j] = merge_step(ws, v[t][u][j], a[t][j]);
          }
        }
    }
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      if (first + t * (int)blockDim.x < n_items) {
        Vec<T, 4> out;
#pragma unroll
        for (int j = 0; j < 4; ++j) out.v[j] = from_f32<T>(merge_finish(a[t][j], tot[gi[t]]));
        stvec<T, 4>(o + (bh0 + gi[t]) * Dv + di[t], out);
      }
    }
  }
  if (threadIdx.x == 0) *ticket = 0u;   // zero again for the next launch (published by the kernel boundary)
}

// Sum over the LPR consecutive lanes that hold one K row.  Up to 16 lanes the exchange is done
// with DPP row operations (no LDS crossbar traffic); wider groups finish with xor-shuffles.
template <int CTRL> __device__ __forceinline__ float dpp_add(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int LPR> __device__ __forceinline__ float group_sum(float x) {
  if (LPR >= 2) x = dpp_add<0xB1>(x);    // quad_perm [1,0,3,2]
  if (LPR >= 4) x = dpp_add<0x4E>(x);    // quad_perm [2,3,0,1]
  if (LPR >= 8) x = dpp_add<0x141>(x);   // row_half_mirror
  if (LPR >= 16) x = dpp_add<0x140>(x);  // row_mirror
  if (LPR >= 32) x += __shfl_xor(x, 16, 64);
  if (LPR >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}

// q . k over one 16-byte vector pair, fp32 accumulate.  16-bit types use the packed dot
// instructions (two MACs per lane-instruction, no widening of K).
__device__ __forceinline__ float dot16(const Vec<bf16_t, 8>& a, const Vec<bf16_t, 8>& b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    d = __builtin_amdgcn_fdot2_f32_bf16(bf2{a.v[2 * j], a.v[2 * j + 1]}, bf2{b.v[2 * j], b.v[2 * j + 1]}, d, false);
  return d;
}
__device__ __forceinline__ float dot16(const Vec<f16_t, 8>& a, const Vec<f16_t, 8>& b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    d = __builtin_amdgcn_fdot2(h2{a.v[2 * j], a.v[2 * j + 1]}, h2{b.v[2 * j], b.v[2 * j + 1]}, d, false);
  return d;
}
__device__ __forceinline__ float dot16(const Vec<float, 4>& a, const Vec<float, 4>& b) {
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) d = fmaf(a.v[j], b.v[j], d);
  return d;
}

// grid: (hkv * (gtotal / G), num_splits, B); block 256 = 4 waves.
// Each wave walks its keys in steps of KPI*U rows.  The loop is software-pipelined: the K/V rows
// of step i+1 and the page-table entries of step i+2 are in flight while step i is computed.
template <typename T, int D, int G, int U, bool CAP>
__global__ __launch_bounds__(256) void decode_attn_kernel(const T* __restrict__ q,
                                                          const T* __restrict__ kbuf,
                                                          const T* __restrict__ vbuf,
                                                          T* __restrict__ o, DecodeParams p) {
  constexpr int VE = V16<T>::n;   // elements per 16-B access
  constexpr int LPR = D / VE;     // lanes per K/V row
  constexpr int KPI = 64 / LPR;   // rows per wave-instruction
  constexpr int KPW = KPI * U;    // rows per wave per step
  static_assert(LPR >= 1 && LPR <= 64 && (LPR & (LPR - 1)) == 0, "head_dim not supported");
  __shared__ float sm_acc[4][G][D];
  __shared__ float sm_m[4][G];
  __shared__ float sm_l[4][G];

  const int gtotal = p.hq / p.hkv;
  const int groups_per_kv = gtotal / G;
  // Workgroups are dealt to the 8 XCDs round-robin by linear id, so with 8 kv heads a plain
  // blockIdx.x -> head map would pin each head (one 256-B slice of every 2-KiB token row) to one
  // XCD and funnel that XCD's traffic through a fraction of its L2 channels.  Rotating the head
  // by (sequence + split) gives every XCD every slice.
  const int xrot = (blockIdx.x + blockIdx.y + blockIdx.z) % gridDim.x;
  const int kvh = xrot / groups_per_kv;
  const int head0 = kvh * gtotal + (xrot % groups_per_kv) * G;
  const int split = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kslot = lane / LPR, dchunk = lane % LPR;

  int64_t page_off, len;
  seq_range(p, b, page_off, len);
  const int64_t chunk = split_chunk(p, len);
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = k0 + chunk < len ? k0 + chunk : len;

  // Scores are (q . k) * sm_scale; without a logit cap the softmax runs in the log2 domain, so
  // log2(e) is folded into the multiplier.
  const float qmul = CAP ? p.sm_scale : p.sm_scale * kLog2e;
  Vec<T, VE> qv[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
    qv[g] = ldvec<T, VE>(q + (int64_t)b * p.q_stride + (int64_t)(head0 + g) * D + dchunk * VE);

  float m[G], l[G], acc[G][VE];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = -INFINITY;
    l[g] = 0.f;
#pragma unroll
    for (int j = 0; j < VE; ++j) acc[g][j] = 0.f;
  }

  const int32_t* pt = p.page_table + page_off;
  const int64_t row_stride = p.k_slot_stride;   // (D == Dv here: the value cache has the same geometry)
  const T* kbase = kbuf + (int64_t)kvh * p.k_head_stride + dchunk * VE;
  const T* vbase = vbuf + (int64_t)kvh * p.v_head_stride + dchunk * VE;

  const int64_t first = k0 + (int64_t)wave * KPW + kslot;   // this lane's row in step 0, u = 0
  const int64_t wave_first = k0 + (int64_t)wave * KPW;
  const int n_iter = wave_first < k1 ? (int)((k1 - wave_first + 4 * KPW - 1) / (4 * KPW)) : 0;
  const int64_t last_key = len - 1;

  // Page-table reads are clamped to the sequence, never predicated: unconditional loads keep the
  // compiler's s_waitcnt counted, so the prefetches below really stay in flight.
  auto load_slots = [&](int it, int32_t (&slot)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t key = first + (int64_t)it * (4 * KPW) + u * KPI;
      key = key < last_key ? key : last_key;
      slot[u] = pt[key];
    }
  };
  auto load_kv = [&](const int32_t (&slot)[U], Vec<T, VE> (&kv)[U], Vec<T, VE> (&vv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) kv[u] = ldvec_nt<T, VE>(kbase + (int64_t)slot[u] * row_stride);
#pragma unroll
    for (int u = 0; u < U; ++u) vv[u] = ldvec_nt<T, VE>(vbase + (int64_t)slot[u] * row_stride);
  };
  auto compute = [&](int it, const Vec<T, VE> (&kv)[U], const Vec<T, VE> (&vv)[U]) {
    float s[U][G];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int g = 0; g < G; ++g) s[u][g] = group_sum<LPR>(dot16(qv[g], kv[u]));
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) valid[u] = first + (int64_t)it * (4 * KPW) + u * KPI < k1;
    float vf[U][VE];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < VE; ++j) vf[u][j] = to_f32(vv[u].v[j]);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float mx = m[g];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float x = s[u][g] * qmul;
        if (CAP) x = p.logit_cap * tanhf(x / p.logit_cap) * kLog2e;
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
        for (int u = 0; u < U; ++u) a = fmaf(pu[u], vf[u][j], a);
        acc[g][j] = a;
      }
    }
  };

  if (n_iter > 0) {
    Vec<T, VE> kvA[U], vvA[U], kvB[U], vvB[U];
    int32_t slotA[U], slotB[U];
    load_slots(0, slotA);
    load_slots(1, slotB);
    load_kv(slotA, kvA, vvA);
    for (int it = 0; it + 1 < n_iter; ++it) {
      load_kv(slotB, kvB, vvB);
      load_slots(it + 2, slotB);
      compute(it, kvA, vvA);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        kvA[u] = kvB[u];
        vvA[u] = vvB[u];
      }
    }
    compute(n_iter - 1, kvA, vvA);
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

// ------------------------------------------------------------------------------------------
// head_dim 128, 16-bit: Q.K^T on the matrix cores, P.V on the vector ALU.
//
// A wave takes 16 keys per step.  K rows are loaded straight into MFMA A-fragments (lane
// (r = l & 15, c = l >> 4) reads 16 B of key r at feature 32*s + 8*c for s = 0..3), the G query
// heads sit in the B operand (columns >= G are zero), so four v_mfma_f32_16x16x32 give the whole
// 16 x G score tile and lane (g, c) ends up with the scores of head g for keys 4c..4c+3 -- with no
// cross-lane reduction at all.  That lane runs the online softmax for its four keys and hands
// p[0..3] and the rescale factor to the 16 lanes of ITS OWN row c through LDS; those lanes hold
// V rows 4c..4c+3 (16 B of each) and the fp32 accumulators for their 8 features.  Every row c
// therefore keeps an independent softmax state; rows and waves are merged once at the end.
// The vector ALU does only the P.V multiply-adds: a wave64 VALU instruction costs 4 cycles, and
// the all-VALU kernel above spends about as long issuing them as the chip needs to stream K/V.
// ------------------------------------------------------------------------------------------
template <typename T, int G, bool CAP>
__global__ __launch_bounds__(256) void decode_attn_mfma_kernel(const T* __restrict__ q,
                                                               const T* __restrict__ kbuf,
                                                               const T* __restrict__ vbuf,
                                                               T* __restrict__ o, DecodeParams p) {
  constexpr int D = 128, VE = 8;
  using frag = typename Mfma<T>::frag;
  __shared__ float sm_acc[4][G][D];
  __shared__ float sm_m[4][G];
  __shared__ float sm_l[4][G];
  __shared__ __attribute__((aligned(16))) float sm_p[4][4][G][8];   // [wave][row c][g]{p0..p3, alpha, m, l, -}

  const int gtotal = p.hq / p.hkv;
  const int groups_per_kv = gtotal / G;
  const int xrot = (blockIdx.x + blockIdx.y + blockIdx.z) % gridDim.x;
  const int kvh = xrot / groups_per_kv;
  const int head0 = kvh * gtotal + (xrot % groups_per_kv) * G;
  const int split = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, c4 = lane >> 4;

  int64_t page_off, len;
  seq_range(p, b, page_off, len);
  const int64_t chunk = split_chunk(p, len);
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = k0 + chunk < len ? k0 + chunk : len;
  const float qmul = CAP ? p.sm_scale : p.sm_scale * kLog2e;

  // B operand: Q[head0 + j][32 s + 8 c + 0..7] in lane (j = r16, c); zero columns for j >= G
  frag qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (r16 < G) {
      qf[s] = *reinterpret_cast<const frag*>(q + (int64_t)b * p.q_stride + (int64_t)(head0 + r16) * D + 32 * s + 8 * c4);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (T)0.f;
    }
  }

  float m = -INFINITY, l = 0.f;   // softmax state of lane (g = r16, row c4); meaningful for r16 < G
  float acc[G][VE];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < VE; ++j) acc[g][j] = 0.f;

  const int32_t* pt = p.page_table + page_off;
  const int64_t row_stride = p.k_slot_stride;
  const T* kbase = kbuf + (int64_t)kvh * p.k_head_stride + 8 * c4;     // + slot * row_stride + 32 s
  const T* vbase = vbuf + (int64_t)kvh * p.v_head_stride + 8 * r16;    // + slot * row_stride
  // A wave takes TL tiles of 16 keys per step (all their K/V loads are issued before the first
  // tile is computed); the 4 waves interleave steps.  Measured: TL = 2 (16 KiB per wave per memory
  // round trip) is no faster than TL = 1 and costs 32 VGPRs.
  constexpr int TL = 1;
  constexpr int STEP = 16 * TL;
  const int64_t wave_first = k0 + (int64_t)wave * STEP;
  const int n_iter = wave_first < k1 ? (int)((k1 - wave_first + 4 * STEP - 1) / (4 * STEP)) : 0;
  const int64_t last_key = len - 1;
  float* my_p = &sm_p[wave][c4][0][0];

  // Page-table reads are clamped to the sequence, never predicated: unconditional loads keep the
  // compiler's s_waitcnt counted.
  auto load_slots = [&](int it, int32_t (&sk)[TL], int32_t (&sv)[TL][4]) {
#pragma unroll
    for (int tl = 0; tl < TL; ++tl) {
      const int64_t tb = wave_first + (int64_t)it * (4 * STEP) + tl * 16;
      int64_t key = tb + r16;
      sk[tl] = pt[key < last_key ? key : last_key];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        key = tb + 4 * c4 + u;
        sv[tl][u] = pt[key < last_key ? key : last_key];
      }
    }
  };
  auto load_kv = [&](const int32_t (&sk)[TL], const int32_t (&sv)[TL][4], frag (&kf)[TL][4],
                     Vec<T, VE> (&vv)[TL][4]) {
#pragma unroll
    for (int tl = 0; tl < TL; ++tl) {
      const T* krow = kbase + (int64_t)sk[tl] * row_stride;
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[tl][s] = *reinterpret_cast<const frag*>(krow + 32 * s);
    }
#pragma unroll
    for (int tl = 0; tl < TL; ++tl)
#pragma unroll
      for (int u = 0; u < 4; ++u) vv[tl][u] = ldvec<T, VE>(vbase + (int64_t)sv[tl][u] * row_stride);
  };

  if (n_iter > 0) {
    // The page-table entries of step it+1 are fetched while step it is computed.  K/V are not
    // double-buffered: with >= 8 waves per CU the other waves' loads cover this wave's
    // arithmetic (measured: a second K/V buffer bought nothing).
    int32_t sk[TL], sv[TL][4];
    frag kf[TL][4];
    Vec<T, VE> vv[TL][4];
    load_slots(0, sk, sv);
    for (int it = 0; it < n_iter; ++it) {
      load_kv(sk, sv, kf, vv);
      load_slots(it + 1, sk, sv);
#pragma unroll
      for (int tl = 0; tl < TL; ++tl) {
        f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) sc = Mfma<T>::run(kf[tl][s], qf[s], sc);

        // lane (g = r16, c4): keys tb + 4 c4 + r
        const int64_t kb0 = wave_first + (int64_t)it * (4 * STEP) + tl * 16 + 4 * c4;
        float x[4], mx = m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sc[r] * qmul;
          if (CAP) v = p.logit_cap * tanhf(v / p.logit_cap) * kLog2e;
          x[r] = kb0 + r < k1 ? v : -INFINITY;
          mx = fmaxf(mx, x[r]);
        }
        const float ms = safe_m(mx);
        const float alpha = exp2f(m - ms);
        f32x4 pr;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = exp2f(x[r] - ms);
          psum += pr[r];
        }
        l = l * alpha + psum;
        m = mx;
        if (r16 < G) {
          *reinterpret_cast<f32x4*>(my_p + r16 * 8) = pr;
          my_p[r16 * 8 + 4] = alpha;
        }
        __builtin_amdgcn_wave_barrier();

        float vf[4][VE];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < VE; ++j) vf[u][j] = to_f32(vv[tl][u].v[j]);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const f32x4 pg = *reinterpret_cast<const f32x4*>(my_p + g * 8);
          const float ag = my_p[g * 8 + 4];
#pragma unroll
          for (int j = 0; j < VE; ++j) {
            float a = acc[g][j] * ag;
#pragma unroll
            for (int u = 0; u < 4; ++u) a = fmaf(pg[u], vf[u][j], a);
            acc[g][j] = a;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }

  // hand the rows' (m, l) to their accumulator lanes
  if (r16 < G) {
    my_p[r16 * 8 + 5] = m;
    my_p[r16 * 8 + 6] = l;
  }
  __builtin_amdgcn_wave_barrier();
  float mg[G], lg[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    mg[g] = my_p[g * 8 + 5];
    lg[g] = my_p[g * 8 + 6];
  }
  // merge the 4 rows of this wave
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float mo = __shfl_xor(mg[g], off, 64), lo = __shfl_xor(lg[g], off, 64);
      const float mx = fmaxf(mg[g], mo), ms = safe_m(mx);
      const float a = exp2f(mg[g] - ms), bb = exp2f(mo - ms);
      lg[g] = lg[g] * a + lo * bb;
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        const float ao = __shfl_xor(acc[g][j], off, 64);
        acc[g][j] = acc[g][j] * a + ao * bb;
      }
      mg[g] = mx;
    }
  }
  if (c4 == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int j = 0; j < VE; ++j) sm_acc[wave][g][r16 * VE + j] = acc[g][j];
      if (r16 == 0) {
        sm_m[wave][g] = mg[g];
        sm_l[wave][g] = lg[g];
      }
    }
  }
  __syncthreads();
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

// ------------------------------------------------------------------------------------------
// head_dim 128, 16-bit: Q.K^T AND P.V on the matrix cores.
//
// Scores as above: lane (g = l & 15, c = l >> 4) gets head g's scores of keys 4c..4c+3 of the
// wave's 16-key tile.  That is exactly the B operand of v_mfma_f32_16x16x16 for O^T = V^T P^T
// (B[k = key 4c + r][j = g]), so the probabilities go from the softmax straight into the next
// MFMA with no cross-lane movement -- rounded to T first, as the reference's decode kernel does
// (p.to(v.dtype) before the dot, triton_ops/decode_attention.py:135-140).  The A operand V^T
// (lane (d, c): keys 4c..4c+3 of feature d) comes from the wave's own row-major LDS copy of the
// V tile through transposed reads.  The softmax maximum is shared by the 16 keys of a tile (two
// cross-lane max steps), which leaves ~80 vector instructions per 16 keys instead of ~250.
// The accumulator holds O^T[d = 16 dt + 4c + r][g]; columns g >= G are padding.
// KROW (the product configuration): K rows are fetched the way V rows are -- the 16 lanes of a
// quarter-wave read ONE whole 256-byte row (two cache lines) per instruction -- and reach MFMA
// fragment order through the wave's own LDS tile.  Loading fragments straight from memory makes
// every quarter-wave touch 16 rows x 16 bytes, i.e. sixteen cache-line look-ups for 256 bytes;
// measured 58.8 -> 56.4 us at batch 64 x 1064 keys, 8 kv heads.
// ------------------------------------------------------------------------------------------
constexpr int DV_ROW = 128 + 16;   // V tile row in LDS (elements): 288 B -> conflict-free transposed reads

//
// FUSED: q, k_new and v_new are summed from the qkv GEMM's split-K partials (split order, one
// rounding -- that GEMM's own combine), q and k_new get neox RoPE with the rounding points of the
// RoPE kernel, the wave that owns the newest key writes k_new / v_new into the cache and uses
// them from registers for that key's tile.  One workgroup per (sequence, kv head, split) is
// required (G = all query heads of the kv head), so nobody else reads or writes that slot.
// KV8: the cache holds e4m3fn bytes (kbuf / vbuf are byte buffers behind the T pointers): rows are
// widened to T on load, the new token's rows are stored quantised and used at their stored value.
template <typename T, int G, bool CAP, bool FUSED = false, bool KV8 = false, bool KROW = false>
__global__ __launch_bounds__(256) void decode_attn_mfma_pv_kernel(const T* __restrict__ q,
                                                                  const T* kbuf, const T* vbuf,
                                                                  T* __restrict__ o, DecodeParams p,
                                                                  FusedQkv fz = FusedQkv()) {
#pragma clang fp contract(off)   // see the note in `compute`
  using KVT = typename std::conditional<KV8, uint8_t, T>::type;
  constexpr int D = 128, VE = 8;
  using frag = typename Mfma<T>::frag;
  using frag4 = typename Mfma16<T>::frag;
  // what the K/V register buffers hold between load and use: with an fp8 cache the raw 8 bytes
  // (widened when the tile is computed), so that two buffered tiles cost half the registers
  using KRaw = typename std::conditional<KV8, uint2, typename std::conditional<KROW, Vec<T, VE>, frag>::type>::type;
  using VRaw = typename std::conditional<KV8, uint2, Vec<T, VE>>::type;
  __shared__ float sm_acc[4][G][D];
  __shared__ float sm_m[4][G];
  __shared__ float sm_l[4][G];
  __shared__ __attribute__((aligned(16))) T sm_v[4][16 * DV_ROW];
  // KROW: K rows are fetched like V rows (a quarter-wave reads one whole 256-byte row, i.e. two
  // cache lines, instead of sixteen 16-byte pieces of sixteen rows) and reach MFMA fragment order
  // through the wave's own LDS tile
  __shared__ __attribute__((aligned(16))) T sm_k[KROW ? 4 : 1][KROW ? 16 * DV_ROW : 8];

  const int gtotal = p.hq / p.hkv;
  const int groups_per_kv = gtotal / G;
  const int xrot = (blockIdx.x + blockIdx.y + blockIdx.z) % gridDim.x;
  const int kvh = xrot / groups_per_kv;
  const int head0 = kvh * gtotal + (xrot % groups_per_kv) * G;
  const int split = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, c4 = lane >> 4;

  int64_t page_off, len;
  seq_range(p, b, page_off, len);
  const int64_t chunk = split_chunk(p, len);
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = k0 + chunk < len ? k0 + chunk : len;
  const float qmul = CAP ? p.sm_scale : p.sm_scale * kLog2e;

  // FUSED: the workgroup first reduces its (G + 2) x 128 slice of the qkv row (q heads, k, v) from
  // the split-K partials into LDS -- one float4 per thread and split -- and applies neox RoPE to the q and
  // k slices on the way (features i and i + 64 of a head exchange through a lane shuffle); the MFMA fragments
  // are then read from LDS as they are.
  __shared__ __attribute__((aligned(16))) T sm_qkv[FUSED ? (G + 2) * D : 8];
  auto reduce_qkv_to_lds = [&]() {
    const int64_t pos = fz.positions[b];
    // Every thread's items (one, or two at G = 8) and up to eight splits go out as ONE batch of loads -- clamped
    // addresses, masked values, never a condition on a load -- so a projection cut eleven ways (a tensor-parallel
    // rank's narrow qkv) costs two memory round trips here, not six.  The sum per element stays in split order.
    constexpr int ITEMS = ((G + 2) * 32 + 255) / 256;
    float sums[ITEMS][4];
    {
      const float* srcs[ITEMS];
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int v = min((int)threadIdx.x + it * 256, (G + 2) * 32 - 1);
        const int seg = v >> 5, f4 = v & 31;
        const int64_t head = seg < G ? head0 + seg : seg == G ? p.hq + kvh : p.hq + p.hkv + kvh;
        srcs[it] = fz.part + (int64_t)b * fz.row + head * D + 4 * f4;
#pragma unroll
        for (int j = 0; j < 4; ++j) sums[it][j] = 0.f;
      }
      if (fz.nsplit <= 4) {
        float4 f[ITEMS][4];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
#pragma unroll
          for (int u = 0; u < 4; ++u) f[it][u] = *reinterpret_cast<const float4*>(srcs[it] + min(u, fz.nsplit - 1) * fz.split_stride);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (u < fz.nsplit) { sums[it][0] += f[it][u].x; sums[it][1] += f[it][u].y; sums[it][2] += f[it][u].z; sums[it][3] += f[it][u].w; }
      } else {
        for (int sp0 = 0; sp0 < fz.nsplit; sp0 += 8) {
          float4 f[ITEMS][8];
#pragma unroll
          for (int it = 0; it < ITEMS; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) f[it][u] = *reinterpret_cast<const float4*>(srcs[it] + min(sp0 + u, fz.nsplit - 1) * fz.split_stride);
#pragma unroll
          for (int it = 0; it < ITEMS; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (sp0 + u < fz.nsplit) { sums[it][0] += f[it][u].x; sums[it][1] += f[it][u].y; sums[it][2] += f[it][u].z; sums[it][3] += f[it][u].w; }
        }
      }
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int v = threadIdx.x + it * 256;
      if (v >= (G + 2) * 32) break;   // (whole 32-lane slices drop out together)
      const int seg = v >> 5, f4 = v & 31;
      const int64_t head = seg < G ? head0 + seg : seg == G ? p.hq + kvh : p.hq + p.hkv + kvh;
      const float (&sum)[4] = sums[it];
      Vec<T, 4> out;
      if (fz.xs) {   // the fp8 GEMM's epilogue: (acc * x_scale[m]) * w_scale[n], cast once
        const float xs = fz.xs[b];
        const float4 w4 = *reinterpret_cast<const float4*>(fz.ws + head * D + 4 * f4);
        out.v[0] = from_f32<T>((sum[0] * xs) * w4.x);
        out.v[1] = from_f32<T>((sum[1] * xs) * w4.y);
        out.v[2] = from_f32<T>((sum[2] * xs) * w4.z);
        out.v[3] = from_f32<T>((sum[3] * xs) * w4.w);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out.v[j] = from_f32<T>(sum[j] + 0.f);
      }
      // neox RoPE on the q heads and the new key, here, once per workgroup (each wave would otherwise rotate
      // its own copy of all heads after the barrier): features i and i + 64 of a head sit in lanes 16 apart of
      // this wave (f4 and f4 + 16).  Rounding points of the RoPE kernel; the value slice is left alone.
      // (All 32 lanes of a slice are active together, and slices never straddle a wave: the exchange is safe.)
      if (seg <= G) {   // (uniform over the 32 lanes of a slice)
#pragma clang fp contract(off)
        const int fl = f4 & 15;
        float c4v[4], s4v[4];
        if (fz.cache_f32) {
          const float4 cc = *reinterpret_cast<const float4*>(static_cast<const float*>(fz.cos_sin) + pos * D + 4 * fl);
          const float4 ss = *reinterpret_cast<const float4*>(static_cast<const float*>(fz.cos_sin) + pos * D + 64 + 4 * fl);
          c4v[0] = cc.x; c4v[1] = cc.y; c4v[2] = cc.z; c4v[3] = cc.w;
          s4v[0] = ss.x; s4v[1] = ss.y; s4v[2] = ss.z; s4v[3] = ss.w;
        } else {
          const Vec<T, 4> cc = ldvec<T, 4>(static_cast<const T*>(fz.cos_sin) + pos * D + 4 * fl);
          const Vec<T, 4> ss = ldvec<T, 4>(static_cast<const T*>(fz.cos_sin) + pos * D + 64 + 4 * fl);
#pragma unroll
          for (int j = 0; j < 4; ++j) { c4v[j] = to_f32(cc.v[j]); s4v[j] = to_f32(ss.v[j]); }
        }
        const bool second = f4 >= 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float mine = to_f32(out.v[j]);
          const float theirs = __shfl_xor(mine, 16, 64);
          const float c = round_to<T>(c4v[j]), sn = round_to<T>(s4v[j]);
          const float x1 = second ? theirs : mine, x2 = second ? mine : theirs;
          const float y = second ? round_to<T>(round_to<T>(x2 * c) + round_to<T>(x1 * sn))
                                 : round_to<T>(round_to<T>(x1 * c) - round_to<T>(x2 * sn));
          out.v[j] = from_f32<T>(y);
        }
      }
      stvec<T, 4>(&sm_qkv[seg * D + 4 * f4], out);
    }
  };
  auto frags_from_lds = [&](int seg, frag (&y)[4]) {   // (rotated already)
#pragma unroll
    for (int s = 0; s < 4; ++s) y[s] = *reinterpret_cast<const frag*>(&sm_qkv[seg * D + 32 * s + 8 * c4]);
  };

  frag qf[4];
  if constexpr (!FUSED) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (r16 < G) {
        qf[s] = *reinterpret_cast<const frag*>(q + (int64_t)b * p.q_stride + (int64_t)(head0 + r16) * D + 32 * s + 8 * c4);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[s][j] = (T)0.f;
      }
    }
  }

  float m = -INFINITY;   // running maximum of head g = r16 (same in the four lanes c of a head)
  float l = 0.f;         // this lane's share of the denominator (its keys 4c..4c+3 of every tile)
  f32x4 acc[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int32_t* pt = p.page_table + page_off;
  const int64_t row_stride = p.k_slot_stride;
  const KVT* kbase = reinterpret_cast<const KVT*>(kbuf) + (int64_t)kvh * p.k_head_stride + 8 * (KROW ? r16 : c4);   // + slot * row_stride (+ 32 s)
  const KVT* vbase = reinterpret_cast<const KVT*>(vbuf) + (int64_t)kvh * p.v_head_stride + 8 * r16;    // + slot * row_stride
  const int64_t wave_first = k0 + (int64_t)wave * 16;
  const int n_iter = wave_first < k1 ? (int)((k1 - wave_first + 63) / 64) : 0;
  const int64_t last_key = len - 1;
  T* v_wr = &sm_v[wave][(4 * c4) * DV_ROW + 8 * r16];                           // + u rows
  T* k_wr = &sm_k[KROW ? wave : 0][KROW ? (4 * c4) * DV_ROW + 8 * r16 : 0];     // + u rows
  const T* k_rd = &sm_k[KROW ? wave : 0][KROW ? r16 * DV_ROW + 8 * c4 : 0];     // + 32 s
  const T* v_rd = &sm_v[wave][(4 * c4 + (r16 >> 2)) * DV_ROW + 4 * (r16 & 3)];  // + 16 dt

  // FUSED: the newest key (position len - 1) belongs to the last split; its 16-key tile is tile
  // `tl` of the split, walked by wave tl & 3 in iteration tl >> 2, row `rl` of the tile.
  frag knew[4];
  Vec<T, VE> vnew;
  bool owner = false;
  int it_new = -1, rl = 0;
  if constexpr (FUSED) {
    if (k1 == len && len > k0) {
      const int tl = (int)((len - 1 - k0) >> 4);
      rl = (int)((len - 1 - k0) & 15);
      it_new = tl >> 2;
      owner = (tl & 3) == wave;
    }
  }

  auto load_slots = [&](int it, int32_t& sk, int32_t (&sv)[4]) {
    const int64_t tb = wave_first + (int64_t)it * 64;
    int64_t key = tb + r16;
    sk = pt[key < last_key ? key : last_key];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      key = tb + 4 * c4 + u;
      sv[u] = pt[key < last_key ? key : last_key];
    }
  };

  auto load_kv = [&](const int32_t& sk, const int32_t (&sv)[4], KRaw (&kf)[4], VRaw (&vv)[4]) {
    const KVT* krow = kbase + (int64_t)sk * row_stride;
    if constexpr (KV8 && KROW) {
#pragma unroll
      for (int u = 0; u < 4; ++u) kf[u] = ld8_nt(kbase + (int64_t)sv[u] * row_stride);
#pragma unroll
      for (int u = 0; u < 4; ++u) vv[u] = ld8_nt(vbase + (int64_t)sv[u] * row_stride);
    } else if constexpr (KV8) {
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const uint2*>(krow + 32 * s);
#pragma unroll
      for (int u = 0; u < 4; ++u) vv[u] = ld8_nt(vbase + (int64_t)sv[u] * row_stride);
    } else if constexpr (KROW) {
#pragma unroll
      for (int u = 0; u < 4; ++u) kf[u] = ldvec_nt<T, VE>(kbase + (int64_t)sv[u] * row_stride);
#pragma unroll
      for (int u = 0; u < 4; ++u) vv[u] = ldvec_nt<T, VE>(vbase + (int64_t)sv[u] * row_stride);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const frag*>(krow + 32 * s);
#pragma unroll
      for (int u = 0; u < 4; ++u) vv[u] = ldvec<T, VE>(vbase + (int64_t)sv[u] * row_stride);
    }
  };
  // One 16-key tile.  A tile past the end of the range is harmless: every key is masked, so the
  // probabilities are zero and the rescale factor is one.
  auto compute_tile = [&](frag (&kf)[4], Vec<T, VE> (&vv)[4], int it) {
    // No multiply-add contraction in the softmax: the fused and the plain instantiation of this
    // kernel must round identically (the compiler otherwise contracts differently per instantiation).
#pragma clang fp contract(off)
    if constexpr (FUSED) {
      if (owner && it == it_new) {   // this tile's row rl was loaded before the cache held it
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (r16 == rl) kf[s] = knew[s];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (4 * c4 + u == rl) vv[u] = vnew;
      }
    }
    f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) sc = Mfma<T>::run(kf[s], qf[s], sc);

    const int64_t kb0 = wave_first + (int64_t)it * 64 + 4 * c4;
    float x[4], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = sc[r] * qmul;
      if (CAP) v = p.logit_cap * tanhf(v / p.logit_cap) * kLog2e;
      x[r] = kb0 + r < k1 ? v : -INFINITY;
      mx = fmaxf(mx, x[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    mx = fmaxf(mx, m);
    const float ms = safe_m(mx);
    const float alpha = exp2f(m - ms);
    float pr[4], psum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr[r] = exp2f(x[r] - ms);
      psum += pr[r];
    }
    l = l * alpha + psum;
    m = mx;
    T pt4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pt4[r] = from_f32<T>(pr[r]);
    const frag4 pb = __builtin_bit_cast(frag4, pt4);

#pragma unroll
    for (int u = 0; u < 4; ++u) stvec<T, VE>(v_wr + u * DV_ROW, vv[u]);
    __builtin_amdgcn_wave_barrier();
    // the running maximum rarely moves after the first tiles: skip the rescale (a multiplication by
    // exactly 1, so skipping it changes no bit) unless some lane of the wave needs it
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) acc[dt] *= alpha;
    }
    // all eight transposed reads first, then the MFMAs: written read-MFMA-read-MFMA the reads stay in
    // that order and every MFMA waits one LDS round trip
    typename TrRead<T>::v4 vt[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) vt[dt] = TrRead<T>::run(v_rd + 16 * dt);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) acc[dt] = Mfma16<T>::run(__builtin_bit_cast(frag4, vt[dt]), pb, acc[dt]);
    __builtin_amdgcn_wave_barrier();
  };

  // widen an fp8 tile on use; a 16-bit tile is computed in place
  auto compute = [&](KRaw (&kraw)[4], VRaw (&vraw)[4], int it) {
    if constexpr (KV8 && KROW) {
      Vec<T, VE> vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) stvec<T, VE>(k_wr + u * DV_ROW, cvt8_from_fp8<T>(kraw[u]));
#pragma unroll
      for (int u = 0; u < 4; ++u) vv[u] = cvt8_from_fp8<T>(vraw[u]);
      __builtin_amdgcn_wave_barrier();
      frag kf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const frag*>(k_rd + 32 * s);
      __builtin_amdgcn_wave_barrier();
      compute_tile(kf, vv, it);
    } else if constexpr (KV8) {
      frag kf[4];
      Vec<T, VE> vv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[s] = __builtin_bit_cast(frag, cvt8_from_fp8<T>(kraw[s]));
#pragma unroll
      for (int u = 0; u < 4; ++u) vv[u] = cvt8_from_fp8<T>(vraw[u]);
      compute_tile(kf, vv, it);
    } else if constexpr (KROW) {
#pragma unroll
      for (int u = 0; u < 4; ++u) stvec<T, VE>(k_wr + u * DV_ROW, kraw[u]);
      __builtin_amdgcn_wave_barrier();
      frag kf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const frag*>(k_rd + 32 * s);
      __builtin_amdgcn_wave_barrier();
      compute_tile(kf, vraw, it);
    } else {
      compute_tile(kraw, vraw, it);
    }
  };

  // Two K/V register buffers: the loads of tile it+1 are in flight while tile it is computed, so a
  // wave keeps 8-16 KiB outstanding at all times.  Addresses past the range are clamped (valid).
  int32_t skA, svA[4], skB, svB[4];
  KRaw kfA[4], kfB[4];
  VRaw vvA[4], vvB[4];
  if (n_iter > 0) {
    load_slots(0, skA, svA);
    load_slots(1, skB, svB);
  }
  if constexpr (FUSED) reduce_qkv_to_lds();   // its loads share the page-table round trip
  if (n_iter > 0) {   // the first two K/V tiles are in flight across the barrier below
    load_kv(skA, svA, kfA, vvA);
    load_kv(skB, svB, kfB, vvB);
    load_slots(2, skA, svA);
    load_slots(3, skB, svB);
  }
  if constexpr (FUSED) {
    __syncthreads();
    frags_from_lds(r16 < G ? r16 : 0, qf);
    if (r16 >= G) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[s][j] = (T)0.f;
    }
    if (owner) {
      frags_from_lds(G, knew);
      vnew = ldvec<T, VE>(&sm_qkv[(G + 1) * D + 8 * r16]);
      const int64_t slot = fz.out_loc[b];
      KVT* kd = const_cast<KVT*>(reinterpret_cast<const KVT*>(kbuf)) + slot * row_stride + (int64_t)kvh * p.k_head_stride + 8 * c4;
      KVT* vd = const_cast<KVT*>(reinterpret_cast<const KVT*>(vbuf)) + slot * row_stride + (int64_t)kvh * p.v_head_stride + 8 * r16;
      if constexpr (KV8) {
        // quantise once; the attention below sees exactly what a later step will read from the cache
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const uint2 qv = quant8_fp8<T>(__builtin_bit_cast(Vec<T, 8>, knew[s]), 1.0f);
          if (r16 == 0) *reinterpret_cast<uint2*>(kd + 32 * s) = qv;
          knew[s] = __builtin_bit_cast(frag, cvt8_from_fp8<T>(qv));
        }
        const uint2 qv = quant8_fp8<T>(vnew, 1.0f);
        if (c4 == 0) *reinterpret_cast<uint2*>(vd) = qv;
        vnew = cvt8_from_fp8<T>(qv);
      } else {
        if (r16 == 0) {
#pragma unroll
          for (int s = 0; s < 4; ++s) *reinterpret_cast<frag*>(kd + 32 * s) = knew[s];
        }
        if (c4 == 0) stvec<T, VE>(vd, vnew);
      }
    }
  }
  if (n_iter > 0) {
    for (int it = 0; it < n_iter; it += 2) {
      compute(kfA, vvA, it);
      load_kv(skA, svA, kfA, vvA);      // tile it + 2
      load_slots(it + 4, skA, svA);
      compute(kfB, vvB, it + 1);
      load_kv(skB, svB, kfB, vvB);      // tile it + 3
      load_slots(it + 5, skB, svB);
    }
  }

  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (r16 < G) {
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sm_acc[wave][r16][16 * dt + 4 * c4 + r] = acc[dt][r];
    if (c4 == 0) {
      sm_m[wave][r16] = m;
      sm_l[wave][r16] = l;
    }
  }
  __syncthreads();
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
      // write-through, so that the workgroup that merges the splits (below) sees them without a fence;
      // the resources start at the workgroup's first head (wave-uniform), the lane adds its head and feature
      const int64_t ph0 = ((int64_t)b * p.hq + head0) * p.num_splits;
      const int cell = g * p.num_splits + split;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val),
                                            wt_rsrc(p.part_o + ph0 * D, G * p.num_splits * D * 4), (cell * D + d) * 4, 0, 16);
      if (d == 0)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lt > 0.f ? mx * kLn2 + logf(lt) : -INFINITY),
                                              wt_rsrc(p.part_lse + ph0, G * p.num_splits * 4), cell * 4, 0, 16);
    }
  }
  // (one counter per sequence and head group: blockIdx.x is rotated, xrot names the group)
  if (p.num_splits > 1 && p.tickets) merge_splits_if_last<T, 1>(p, o, b, head0, G, D, b * (int)gridDim.x + xrot, &sm_acc[0][0][0]);
}

// Any head_dim / v_head_dim (<= 1024): one wave per (head, split, sequence); lanes stride the
// feature axis.  Slow path for shapes outside the vectorised kernel (e.g. D = 80, 96, 576/512).
template <typename T>
__global__ __launch_bounds__(64) void decode_attn_generic_kernel(const T* __restrict__ q,
                                                                 const T* __restrict__ kbuf,
                                                                 const T* __restrict__ vbuf,
                                                                 T* __restrict__ o, DecodeParams p,
                                                                 int D, int Dv, int kv8 = 0) {
  // kv8: the cache holds e4m3fn bytes behind the T pointers (strides count bytes); widened exactly on load
  const int head = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x;
  const int kvh = head / (p.hq / p.hkv);
  int64_t page_off, len;
  seq_range(p, b, page_off, len);
  const int64_t chunk = split_chunk(p, len);
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = k0 + chunk < len ? k0 + chunk : len;
  const T* qh = q + (int64_t)b * p.q_stride + (int64_t)head * D;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int64_t key = k0; key < k1; ++key) {
    const int64_t slot = p.page_table[page_off + key];
    const int64_t koff = slot * p.k_slot_stride + (int64_t)kvh * p.k_head_stride;
    const int64_t voff = slot * p.v_slot_stride + (int64_t)kvh * p.v_head_stride;
    const T* kr = kbuf + koff;
    const T* vr = vbuf + voff;
    const uint8_t* kr8 = reinterpret_cast<const uint8_t*>(kbuf) + koff;
    const uint8_t* vr8 = reinterpret_cast<const uint8_t*>(vbuf) + voff;
    float d = 0.f;
    for (int i = lane; i < D; i += 64)
      d = fmaf(to_f32(qh[i]), kv8 ? __builtin_amdgcn_cvt_f32_fp8((int)kr8[i], 0) : to_f32(kr[i]), d);
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
      if (dd < Dv) acc[i] = acc[i] * alpha + pe * (kv8 ? __builtin_amdgcn_cvt_f32_fp8((int)vr8[dd], 0) : to_f32(vr[dd]));
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

// ------------------------------------------------------------------------------------------
// MLA decode (DeepSeek absorbed form): every query head attends over ONE shared latent row per
// token, [kv_lora_rank 512 | rope 64] = 576 features; the value is its first 512 features.
// Sixteen heads are exactly one MFMA tile: a workgroup takes 16 heads of one sequence (and one KV
// split), so each 1152-byte row is read from HBM once per 16 heads and used for both products.
//   scores: lane (key r = l & 15, c = l >> 4) loads the row straight into 18 A-fragments; the heads'
//           queries sit in B (18 fragments, held for the whole kernel); 18 v_mfma_f32_16x16x32.
//   values: the same fragments (features < 512) are written to the wave's row-major LDS tile; V^T
//           comes back through transposed reads, the probabilities are already the B operand of
//           v_mfma_f32_16x16x16 (as in the kernel above); 32 feature tiles -> O^T[512][16 heads].
// Register budget ~400 per lane (2 x 18 row fragments, 18 query fragments, 128 accumulators): one
// workgroup per CU, which still keeps 4 waves x 36 KiB of row loads in flight.
// grid (ceil(hq / 16), num_splits, B), block 256.
// ------------------------------------------------------------------------------------------
constexpr int MLA_RANK = 512, MLA_ROPE = 64, MLA_D = MLA_RANK + MLA_ROPE;
constexpr int MLA_ROW = MLA_RANK + 16;   // LDS row (elements): 1056 B, same bank residue as DV_ROW

constexpr int MLA_QROW = MLA_D + 16;                  // query tile row in LDS: 1184 B (bank = 8 * row residue class, as DV_ROW)
constexpr int MLA_V_BYTES = 4 * 16 * MLA_ROW * 2;     // four waves' value tiles (16-bit elements)
constexpr int MLA_LDS_BYTES = MLA_V_BYTES + 16 * MLA_QROW * 2;
static_assert(MLA_V_BYTES >= 16 * MLA_RANK * 4, "the merge buffer reuses the value tiles");

// KV8: the latent cache holds OCP e4m3fn bytes (576 per row).  A lane then fetches 8 bytes per
// fragment instead of 16, keeps the two buffered tiles as raw bytes (half the registers) and widens a
// tile to T -- exactly, every e4m3 value is a T value -- when it is computed: same results as this
// kernel over a 16-bit cache that holds the widened values.
template <typename T, bool CAP, bool KV8 = false>
__global__ __launch_bounds__(256) void decode_attn_mla_mfma_kernel(const T* __restrict__ q,
                                                                   const T* __restrict__ kvbuf,
                                                                   T* __restrict__ o, DecodeParams p) {
  constexpr int NS = MLA_D / 32;      // 18 score k-steps
  constexpr int NDT = MLA_RANK / 16;  // 32 value feature tiles
  using frag = typename Mfma<T>::frag;
  using frag4 = typename Mfma16<T>::frag;
  using rawfrag = typename std::conditional<KV8, uint2, frag>::type;   // one lane's 8 features of a row as loaded
  // the value tiles during the key loop; afterwards the cross-wave accumulator [16 heads][512]
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];   // MLA_LDS_BYTES (> 64 KiB: dynamic)
  __shared__ float sm_m[4][16];
  __shared__ float sm_l[4][16];

  const int head0 = blockIdx.x * 16, split = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, c4 = lane >> 4;
  const bool head_ok = head0 + r16 < p.hq;

  int64_t page_off, len;
  seq_range(p, b, page_off, len);
  const int64_t chunk = split_chunk(p, len);
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = k0 + chunk < len ? k0 + chunk : len;
  const float qmul = CAP ? p.sm_scale : p.sm_scale * kLog2e;

  // The 16 heads' queries live in LDS (rows padded to 1168 B) and are re-read per tile: 18 more
  // fragments in registers would not fit beside two row buffers and the accumulators.
  T* sm_q = reinterpret_cast<T*>(sm_raw + MLA_V_BYTES);
  for (int i = threadIdx.x; i < 16 * (MLA_D / 8); i += 256) {
    const int h = i / (MLA_D / 8), c = i % (MLA_D / 8);
    Vec<T, 8> v;
    if (head0 + h < p.hq) {
      v = ldvec<T, 8>(q + (int64_t)b * p.q_stride + (int64_t)(head0 + h) * MLA_D + 8 * c);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v.v[j] = from_f32<T>(0.f);
    }
    stvec<T, 8>(sm_q + h * MLA_QROW + 8 * c, v);
  }
  __syncthreads();
  const T* q_rd = sm_q + r16 * MLA_QROW + 8 * c4;   // + 32 s

  float m = -INFINITY, l = 0.f;
  f32x4 acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int32_t* pt = p.page_table + page_off;
  const T* kbase = kvbuf + 8 * c4;                        // + slot * k_slot_stride + 32 s
  const int64_t wave_first = k0 + (int64_t)wave * 16;
  const int n_iter = wave_first < k1 ? (int)((k1 - wave_first + 63) / 64) : 0;
  const int64_t last_key = len - 1;
  T* my_v = reinterpret_cast<T*>(sm_raw) + wave * (16 * MLA_ROW);
  T* v_wr = my_v + r16 * MLA_ROW + 8 * c4;                                  // + 32 s
  const T* v_rd = my_v + (4 * c4 + (r16 >> 2)) * MLA_ROW + 4 * (r16 & 3);   // + 16 dt

  auto load_slot = [&](int it) -> int32_t {
    const int64_t key = wave_first + (int64_t)it * 64 + r16;
    return pt[key < last_key ? key : last_key];
  };
  auto load_row = [&](int32_t slot, rawfrag (&kf)[NS]) {
    if constexpr (KV8) {
      const uint8_t* krow = reinterpret_cast<const uint8_t*>(kvbuf) + 8 * c4 + (int64_t)slot * p.k_slot_stride;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(krow + 32 * s));
        kf[s] = uint2{v[0], v[1]};
      }
    } else {
      const T* krow = kbase + (int64_t)slot * p.k_slot_stride;
#pragma unroll
      for (int s = 0; s < NS; ++s) kf[s] = *reinterpret_cast<const frag*>(krow + 32 * s);
    }
  };
  auto compute = [&](const rawfrag (&kraw)[NS], int it) {
    frag kf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if constexpr (KV8) kf[s] = __builtin_bit_cast(frag, cvt8_from_fp8<T>(kraw[s]));
      else kf[s] = kraw[s];
    }
    f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) sc = Mfma<T>::run(kf[s], *reinterpret_cast<const frag*>(q_rd + 32 * s), sc);
    const int64_t kb0 = wave_first + (int64_t)it * 64 + 4 * c4;
    float x[4], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = sc[r] * qmul;
      if (CAP) v = p.logit_cap * tanhf(v / p.logit_cap) * kLog2e;
      x[r] = kb0 + r < k1 ? v : -INFINITY;
      mx = fmaxf(mx, x[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // Lazy rescale: the reference point m moves only when a tile's maximum exceeds it by more than
    // 2^8 (probabilities then stay below 256 -- exact in fp32, same relative rounding in 16 bits),
    // so the 128 accumulator registers are rescaled a handful of times per sequence, not per tile.
    const bool need = mx > m + 8.f;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
      const float mnew = need ? mx : m;
      const float alpha = exp2f(m - safe_m(mnew));
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) acc[dt] *= alpha;
      m = mnew;
    }
    const float ms = safe_m(m);
    float pr[4], psum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr[r] = exp2f(x[r] - ms);
      psum += pr[r];
    }
    l += psum;
    T pt4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pt4[r] = from_f32<T>(pr[r]);
    const frag4 pb = __builtin_bit_cast(frag4, pt4);
#pragma unroll
    for (int s = 0; s < MLA_RANK / 32; ++s) *reinterpret_cast<frag*>(v_wr + 32 * s) = kf[s];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const auto va = TrRead<T>::run(v_rd + 16 * dt);
      acc[dt] = Mfma16<T>::run(__builtin_bit_cast(frag4, va), pb, acc[dt]);
    }
    __builtin_amdgcn_wave_barrier();
  };

  if (n_iter > 0) {
    rawfrag kfA[NS], kfB[NS];
    int32_t sA = load_slot(0), sB = load_slot(1);
    load_row(sA, kfA);
    load_row(sB, kfB);
    sA = load_slot(2);
    sB = load_slot(3);
    for (int it = 0; it < n_iter; it += 2) {
      compute(kfA, it);
      load_row(sA, kfA);
      sA = load_slot(it + 4);
      compute(kfB, it + 1);
      load_row(sB, kfB);
      sB = load_slot(it + 5);
    }
  }

  // merge the four waves: common maximum per head, then the waves add their rescaled accumulators
  // into one fp32 tile in turn (fixed order: deterministic)
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (c4 == 0) {
    sm_m[wave][r16] = m;
    sm_l[wave][r16] = l;
  }
  __syncthreads();   // every wave is done with its value tile: sm_raw becomes the merge buffer
  float mg = -INFINITY;
#pragma unroll
  for (int w = 0; w < 4; ++w) mg = fmaxf(mg, sm_m[w][r16]);
  const float msg = safe_m(mg);
  const float mine = exp2f(m - msg);
  float lt = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) lt += sm_l[w][r16] * exp2f(sm_m[w][r16] - msg);
  float* macc = reinterpret_cast<float*>(sm_raw);   // [16 heads][512]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        float* dst = macc + r16 * MLA_RANK + 16 * dt + 4 * c4;
        f32x4 v = acc[dt] * mine;
        if (w > 0) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
    __syncthreads();
  }
  const float inv = lt > 0.f ? 1.f / lt : 0.f;
  if (head_ok) {
    const int head = head0 + r16;
    // lane (head r16, c4) writes features [128 c4 + 32 wave, + 32) of its head
    const float* src = macc + r16 * MLA_RANK + 128 * c4 + 32 * wave;
    if (p.num_splits == 1) {
      T* dst = o + ((int64_t)b * p.hq + head) * MLA_RANK + 128 * c4 + 32 * wave;
#pragma unroll
      for (int i = 0; i < 32; ++i) dst[i] = from_f32<T>(lt > 0.f ? src[i] / lt : 0.f);
    } else {
      const int64_t ph = ((int64_t)b * p.hq + head) * p.num_splits + split;
      float* dst = p.part_o + ph * MLA_RANK + 128 * c4 + 32 * wave;
#pragma unroll
      for (int i = 0; i < 32; ++i) dst[i] = lt > 0.f ? src[i] / lt : 0.f;
      if (c4 == 0 && wave == 0) p.part_lse[ph] = lt > 0.f ? mg * kLn2 + logf(lt) : -INFINITY;
    }
  }
  (void)inv;
}

// ------------------------------------------------------------------------------------------
// MLA decode, latent rows landed in LDS by the load unit itself (global_load_lds_dwordx4: no
// register destination).  The register version above keeps two 16-key tiles in 144 registers next
// to 128 accumulators: one workgroup per CU, and everything a workgroup does before its first row
// arrives (sequence bounds -> page table -> rows: three dependent trips to memory) or after its
// last one (the four waves' merge) leaves the CU's memory pipe idle, 0.47 of the HBM peak.  Here:
//   * a ring of 4 tiles x 16 keys in LDS; a step consumes two tiles while the next two (36 KiB per
//     workgroup) are in flight; 74 KiB of LDS and 150 registers -> two workgroups per CU, so one
//     workgroup's start-up and steps overlap the other's loads;
//   * all four waves compute the (cheap) 32 x 16 scores of a step redundantly, so they agree on the
//     softmax statistics bit for bit, and each accumulates ONE quarter of the 512 value features:
//     32 accumulator registers per lane, the 18 query fragments stay in registers, and there is no
//     cross-wave merge at the end;
//   * page-table entries come through the scalar unit (they are wave-uniform: a 1-KiB load
//     instruction fetches the first 1024 bytes of ONE row, addressed as scalar base + 16 * lane), so
//     the vector-memory counter counts nothing but tile loads.
// LDS image of a tile: row r's 512 latent features at 1056 r (1024 + 32 bytes of padding: the
// A-fragment reads (ds_read_b128) and the transposed value reads are both conflict-free), then the
// rows' 64 rope features as [16][128 bytes] with the 16-byte chunk index XOR-ed by (r >> 1) & 7 on
// the SOURCE address and again on the read (the load unit writes lane-linear).
// A wave issues 5 loads per tile: rows 4 w .. 4 w + 3 (1 KiB each) and their rope parts (half a wave).
// grid (ceil(hq / 16), num_splits, B), block 256.
// ------------------------------------------------------------------------------------------
// 0 = plan (two teams per workgroup when the grid allows), 1 = one team (ablation; development build, mi355_debug_set key 10)
#ifdef MI355_DEV_BUILD
int g_mla_teams = 0;
#else
static constexpr int g_mla_teams = 0;
#endif

constexpr int MLR_ROWB = 1056;                       // bytes between rows' latent parts
constexpr int MLR_TAIL = 16 * MLR_ROWB;              // byte offset of the rope parts
constexpr int MLR_TILE = MLR_TAIL + 16 * 128;        // 18944
// NSB step buffers of two tiles each: 2 -> 75776 bytes (two workgroups per CU, one step in flight),
// 4 -> 151552 bytes (one workgroup per CU, three steps = 111 KiB in flight)
constexpr int mlr_lds_bytes(int nsb) { return 2 * nsb * MLR_TILE; }

__device__ __forceinline__ unsigned lds_byte_address(const void* p) {
  return (unsigned)(uintptr_t)(const char __attribute__((address_space(3)))*)(p);
}

// One wave-wide load of 16 bytes per active lane to LDS at `lds_base` (wave-uniform) + 16 * lane.
// M0 carries the LDS base; it is saved and restored inside the statement.
//   _v: per-lane 64-bit source address; _s: wave-uniform 64-bit base + per-lane 32-bit byte offset.
// NT: the non-temporal cache policy, for rows that are read once per step (decode); rows that several
// workgroups share within a launch (extend) keep the default policy.
template <bool NT = false>
__device__ __forceinline__ void load_to_lds_16_v(const void* src, unsigned lds_base) {
  unsigned keep;
  if constexpr (NT)
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_base)
        : "memory");
  else
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_base)
        : "memory");
}
template <bool NT = false>
__device__ __forceinline__ void load_to_lds_16_s(uint64_t base, unsigned lane_off, unsigned lds_base) {
  unsigned keep;
  if constexpr (NT)
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(base), "s"(lds_base)
        : "memory");
  else
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(base), "s"(lds_base)
        : "memory");
}

// Inputs of the decode launch that also applies RoPE and writes the new token's latent row: q's rope part
// and k_pe arrive UNROTATED, the new row [k_nope | rope(k_pe)] is not in the cache yet.
struct MlaRope {
  const void* k_nope;         // [B] rows of 512 features (the normed latent), row stride below (elements)
  const void* k_pe;           // [B] rows of 64 features, not rotated
  int64_t k_nope_stride, k_pe_stride;
  const int64_t* positions;   // [B]
  const int64_t* out_loc;     // [B] cache slot of the new token (padded rows of a captured batch name the sink slot)
  const void* cos_sin;        // [max_pos][64]: cos (32) | sin (32), fp32 or the tensor dtype
  int cache_f32, is_neox;
};

template <typename T> __device__ __forceinline__ float mla_cs_t(const MlaRope& rp, int64_t pos, int idx) {
  return rp.cache_f32 ? static_cast<const float*>(rp.cos_sin)[pos * MLA_ROPE + idx]
                      : to_f32(static_cast<const T*>(rp.cos_sin)[pos * MLA_ROPE + idx]);
}

// ROPE: the launch rotates the query's rope part (in registers) and the new token's k_pe, writes the new
// latent row to the cache (one workgroup per sequence) and -- in every workgroup whose key range ends at the
// new token -- puts that row into the LDS tile over whatever the load unit fetched from the not yet written
// slot.  Same bits as mi355_rope_inplace + mi355_set_kv_buffer + this kernel without ROPE.
// TEAMS = 2: two teams of four waves per workgroup, each with its own ring, on two consecutive KV splits of the
// launch's plan; their results meet in LDS and ONE partial leaves the workgroup (see the fp8 ring kernel below:
// what a launch pays for its splits is the merge of S partials of 16 heads x 512 features).  grid.y = num_splits / TEAMS.
template <typename T, bool CAP, int NSB, bool ROPE = false, int TEAMS = 1>
__global__ __launch_bounds__(256 * TEAMS) void decode_attn_mla_ring_kernel(
    const T* __restrict__ q, const T* __restrict__ kvbuf, T* __restrict__ o, DecodeParams p, MlaRope rp) {
  constexpr int NS = MLA_D / 32;   // 18 score k-steps
  using frag = typename Mfma<T>::frag;
  using frag4 = typename Mfma16<T>::frag;
  constexpr int P = NSB - 1;   // steps requested ahead of the one being computed
  static_assert(P >= 1 && P <= 3, "the wait below knows 0, 10 or 20 younger loads");
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];   // TEAMS * mlr_lds_bytes(NSB)

  const int head0 = blockIdx.x * 16, b = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int team = TEAMS > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;
  const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);   // within the team
  const int tid = threadIdx.x & 255;                                         // within the team
  const int split = blockIdx.y * TEAMS + team;
  char* sm_team = sm_raw + team * mlr_lds_bytes(NSB);
  const int r16 = lane & 15, c4 = lane >> 4;

  int64_t page_off, len;
  seq_range(p, b, page_off, len);
  const int64_t chunk = split_chunk(p, len);
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = k0 + chunk < len ? k0 + chunk : len;
  const int n_steps = k1 > k0 ? (int)((k1 - k0 + 31) / 32) : 0;   // 32 keys (two tiles) per step
  const float qmul = CAP ? p.sm_scale : p.sm_scale * kLog2e;
  // every team's step count and whether its range ends at the new token: the barriers are shared
  int n_team[TEAMS], n_all = 0;
  bool new_team[TEAMS];
#pragma unroll
  for (int tt = 0; tt < TEAMS; ++tt) {
    const int64_t a0 = (int64_t)(blockIdx.y * TEAMS + tt) * chunk;
    const int64_t a1 = a0 + chunk < len ? a0 + chunk : len;
    n_team[tt] = a1 > a0 ? (int)((a1 - a0 + 31) / 32) : 0;
    new_team[tt] = ROPE && a1 == len && len > a0;
    n_all = n_team[tt] > n_all ? n_team[tt] : n_all;
  }

  // ROPE: threads 0..63 hold one 16-byte piece of the new row's latent part, threads 64..71 (interleaved
  // pairs) or 64..67 (neox pairs: a piece and its partner 32 features on) its rotated rope part.
  const bool has_new = ROPE && k1 == len && len > k0;   // this workgroup's range ends at the new token
  Vec<T, 8> new_a, new_b;
  int64_t rope_pos = 0;
  if constexpr (ROPE) {
    rope_pos = rp.positions[b];
    if (has_new) {
      if (tid < 64) {
        new_a = ldvec<T, 8>(static_cast<const T*>(rp.k_nope) + (int64_t)b * rp.k_nope_stride + 8 * tid);
      } else if (tid < 72) {
        const T* kp = static_cast<const T*>(rp.k_pe) + (int64_t)b * rp.k_pe_stride;
        const int c = tid - 64;
        if (!rp.is_neox) {
          const Vec<T, 8> x = ldvec<T, 8>(kp + 8 * c);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = 4 * c + j;
            rope_pair<T>(to_f32(x.v[2 * j]), to_f32(x.v[2 * j + 1]), mla_cs_t<T>(rp, rope_pos, i),
                         mla_cs_t<T>(rp, rope_pos, MLA_ROPE / 2 + i), new_a.v[2 * j], new_a.v[2 * j + 1]);
          }
        } else if (c < 4) {
          const Vec<T, 8> x1 = ldvec<T, 8>(kp + 8 * c), x2 = ldvec<T, 8>(kp + MLA_ROPE / 2 + 8 * c);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int i = 8 * c + j;
            rope_pair<T>(to_f32(x1.v[j]), to_f32(x2.v[j]), mla_cs_t<T>(rp, rope_pos, i),
                         mla_cs_t<T>(rp, rope_pos, MLA_ROPE / 2 + i), new_a.v[j], new_b.v[j]);
          }
        }
      }
      if (blockIdx.x == 0) {   // one workgroup per sequence writes the row to the cache
        T* dst = const_cast<T*>(kvbuf) + rp.out_loc[b] * p.k_slot_stride;
        if (tid < 64) stvec<T, 8>(dst + 8 * tid, new_a);
        else if (tid < 72 && !rp.is_neox) stvec<T, 8>(dst + MLA_RANK + 8 * (tid - 64), new_a);
        else if (tid < 68) {
          stvec<T, 8>(dst + MLA_RANK + 8 * (tid - 64), new_a);
          stvec<T, 8>(dst + MLA_RANK + MLA_ROPE / 2 + 8 * (tid - 64), new_b);
        }
      }
    }
  }

  // page-table reads through the scalar unit (wave-uniform index, read-only data)
  const int32_t __attribute__((address_space(4)))* pt =
      (const int32_t __attribute__((address_space(4)))*)(p.page_table + page_off);
  const int64_t last_key = len - 1;
  struct Slots { int32_t a0, a1, a2, a3, b0, b1, b2, b3; };   // rows 4 wave .. + 3 of a step's two tiles
  auto slot_of = [&](int64_t key) -> int32_t { return pt[key < last_key ? key : last_key]; };
  auto slots_of = [&](int step) -> Slots {
    const int64_t key = k0 + (int64_t)step * 32 + 4 * wave;
    return Slots{slot_of(key), slot_of(key + 1), slot_of(key + 2), slot_of(key + 3),
                 slot_of(key + 16), slot_of(key + 17), slot_of(key + 18), slot_of(key + 19)};
  };
  const unsigned ring = lds_byte_address(sm_team);
  const uint64_t kv_base = (uint64_t)(uintptr_t)kvbuf;
  const uint64_t row_bytes = (uint64_t)p.k_slot_stride * sizeof(T);
  const unsigned lane16 = 16u * lane;
  const int trow = lane >> 3, tch = lane & 7;   // rope part: lane (row 4 wave + trow (lanes 0..31), chunk tch)
  const unsigned tsw = MLA_RANK * sizeof(T) + 16 * (tch ^ (((4 * wave + trow) >> 1) & 7));   // byte offset of the chunk this lane fetches
  const int tm0 = -(int)(trow == 0), tm1 = -(int)(trow == 1), tm2 = -(int)(trow == 2), tm3 = -(int)(trow == 3);
  auto issue_tile = [&](int buf, int32_t s0, int32_t s1, int32_t s2, int32_t s3) {
    const unsigned base = ring + (unsigned)buf * MLR_TILE + (unsigned)(4 * wave) * MLR_ROWB;
    load_to_lds_16_s<true>(kv_base + (uint64_t)(uint32_t)s0 * row_bytes, lane16, base);
    load_to_lds_16_s<true>(kv_base + (uint64_t)(uint32_t)s1 * row_bytes, lane16, base + MLR_ROWB);
    load_to_lds_16_s<true>(kv_base + (uint64_t)(uint32_t)s2 * row_bytes, lane16, base + 2 * MLR_ROWB);
    load_to_lds_16_s<true>(kv_base + (uint64_t)(uint32_t)s3 * row_bytes, lane16, base + 3 * MLR_ROWB);
    // row trow's slot by masks (a chain of selects would be turned into an indexed array in scratch memory)
    const uint32_t ts = (uint32_t)((s0 & tm0) | (s1 & tm1) | (s2 & tm2) | (s3 & tm3));
    if (lane < 32)
      load_to_lds_16_v<true>(reinterpret_cast<const char*>(kvbuf) + (uint64_t)ts * row_bytes + tsw,
                       ring + (unsigned)buf * MLR_TILE + MLR_TAIL + (unsigned)(4 * wave) * 128);
  };
  auto issue_step = [&](int step, const Slots& sl) {
    const int buf = 2 * (step % NSB);
    issue_tile(buf, sl.a0, sl.a1, sl.a2, sl.a3);
    issue_tile(buf + 1, sl.b0, sl.b1, sl.b2, sl.b3);
  };

  float m = -INFINITY, l = 0.f;
  f32x4 acc[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // read addresses inside a tile (bytes)
  const int k_rd = MLR_ROWB * r16 + 16 * c4;                                          // + 64 s
  const int kt_rd0 = MLR_TAIL + 128 * r16 + 16 * (c4 ^ ((r16 >> 1) & 7));             // rope k-step 16
  const int kt_rd1 = MLR_TAIL + 128 * r16 + 16 * ((4 + c4) ^ ((r16 >> 1) & 7));       // rope k-step 17
  const int v_rd = MLR_ROWB * (4 * c4 + (r16 >> 2)) + 8 * (r16 & 3) + 256 * wave;    // + 32 dt

  Slots s_nxt{};
  if (n_steps > 0) {   // (an empty sequence has no page-table entry to read)
#pragma unroll
    for (int i = 0; i < P; ++i) {
      s_nxt = slots_of(i);   // clamped to the last key when past the end
      if (i < n_steps) issue_step(i, s_nxt);
    }
    s_nxt = slots_of(P);
  }

  // B operand of the scores: this lane's head (r16), features 32 s + 8 c4 .. + 7.  Fetched AFTER the first
  // tiles were requested, and waited for here, before the loop: the compiler does not count the
  // loads-to-LDS, and a wait for these fragments placed inside the loop would drain the ring every step.
  frag qf[NS];
  {
    const int hq_c = head0 + r16 < p.hq ? head0 + r16 : p.hq - 1;   // rows of heads past hq are computed and dropped
    const T* qp = q + (int64_t)b * p.q_stride + (int64_t)hq_c * MLA_D + 8 * c4;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const frag*>(qp + 32 * s);
    if constexpr (ROPE) {
      // rope part = k-steps 16 and 17: the lane holds features 8 c4 .. + 7 and 32 + 8 c4 .. + 7 of its head
      Vec<T, 8> a = __builtin_bit_cast(Vec<T, 8>, qf[16]), b2 = __builtin_bit_cast(Vec<T, 8>, qf[17]);
      if (!rp.is_neox) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ia = 4 * c4 + j, ib = 16 + 4 * c4 + j;
          rope_pair<T>(to_f32(a.v[2 * j]), to_f32(a.v[2 * j + 1]), mla_cs_t<T>(rp, rope_pos, ia),
                       mla_cs_t<T>(rp, rope_pos, MLA_ROPE / 2 + ia), a.v[2 * j], a.v[2 * j + 1]);
          rope_pair<T>(to_f32(b2.v[2 * j]), to_f32(b2.v[2 * j + 1]), mla_cs_t<T>(rp, rope_pos, ib),
                       mla_cs_t<T>(rp, rope_pos, MLA_ROPE / 2 + ib), b2.v[2 * j], b2.v[2 * j + 1]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = 8 * c4 + j;
          rope_pair<T>(to_f32(a.v[j]), to_f32(b2.v[j]), mla_cs_t<T>(rp, rope_pos, i),
                       mla_cs_t<T>(rp, rope_pos, MLA_ROPE / 2 + i), a.v[j], b2.v[j]);
        }
      }
      qf[16] = __builtin_bit_cast(frag, a);
      qf[17] = __builtin_bit_cast(frag, b2);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" ::"v"(qf[s]));
  }

  // one step (32 keys = two tiles at `tile`): scores, online softmax, this wave's quarter of P.V
  auto step_compute = [&](int t, const char* tile) {
    f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = f32x4{0.f, 0.f, 0.f, 0.f};
    // key fragments are read MLR_AHEAD k-steps before the MFMAs that use them (a ring of registers; the
    // fences keep the order): left alone the compiler puts each read next to its MFMA, and the score
    // phase then runs at one LDS round trip per k-step instead of at the matrix rate
    constexpr int MLR_AHEAD = 4;
    auto k_off = [&](int s) { return s < 16 ? k_rd + 64 * s : (s == 16 ? kt_rd0 : kt_rd1); };
    frag ka[MLR_AHEAD], kb[MLR_AHEAD];
#pragma unroll
    for (int s = 0; s < MLR_AHEAD; ++s) {
      ka[s] = *reinterpret_cast<const frag*>(tile + k_off(s));
      kb[s] = *reinterpret_cast<const frag*>(tile + MLR_TILE + k_off(s));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      sa = Mfma<T>::run(ka[s % MLR_AHEAD], qf[s], sa);
      sb = Mfma<T>::run(kb[s % MLR_AHEAD], qf[s], sb);
      __builtin_amdgcn_sched_barrier(0);
      if (s + MLR_AHEAD < NS) {
        ka[s % MLR_AHEAD] = *reinterpret_cast<const frag*>(tile + k_off(s + MLR_AHEAD));
        kb[s % MLR_AHEAD] = *reinterpret_cast<const frag*>(tile + MLR_TILE + k_off(s + MLR_AHEAD));
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    const int64_t kb0 = k0 + (int64_t)t * 32 + 4 * c4;
    float x[8], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float v = (r < 4 ? sa[r & 3] : sb[r & 3]) * qmul;
      if (CAP) v = p.logit_cap * tanhf(v / p.logit_cap) * kLog2e;
      x[r] = kb0 + (r & 3) + 16 * (r >> 2) < k1 ? v : -INFINITY;
      mx = fmaxf(mx, x[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // lazy rescale, as in the register kernel: the reference point moves only when a step's maximum exceeds it by 2^8
    const bool need = mx > m + 8.f;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
      const float mnew = need ? mx : m;
      const float alpha = exp2f(m - safe_m(mnew));
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) acc[dt] *= alpha;
      m = mnew;
    }
    const float ms = safe_m(m);
    float psum = 0.f;
    T pa4[4], pb4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pa = exp2f(x[r] - ms), pb = exp2f(x[4 + r] - ms);
      psum += pa + pb;
      pa4[r] = from_f32<T>(pa);
      pb4[r] = from_f32<T>(pb);
    }
    l += psum;
    const frag4 pa = __builtin_bit_cast(frag4, pa4), pb = __builtin_bit_cast(frag4, pb4);
    // transposed value reads in two batches of eight ahead of their MFMAs (see the score phase above)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      typename TrRead<T>::v4 va[4], vb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va[i] = TrRead<T>::run(reinterpret_cast<const T*>(tile + v_rd + 32 * (4 * h + i)));
        vb[i] = TrRead<T>::run(reinterpret_cast<const T*>(tile + MLR_TILE + v_rd + 32 * (4 * h + i)));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[4 * h + i] = Mfma16<T>::run(__builtin_bit_cast(frag4, va[i]), pa, acc[4 * h + i]);
        acc[4 * h + i] = Mfma16<T>::run(__builtin_bit_cast(frag4, vb[i]), pb, acc[4 * h + i]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  {
    for (int t = 0; t < n_all; ++t) {
      const bool live = t < n_steps;   // (team-uniform; a team that has run out of steps keeps the barriers company)
      if (live) {
        // this wave's part of step t has landed when at most the loads (10 per step) of the younger steps are out
        const int younger = min(P - 1, n_steps - 1 - t);
        if (P >= 3 && younger >= 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (P >= 2 && younger == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // and its reads of step t - 1 have returned
      __builtin_amdgcn_s_barrier();                         // every wave's part of step t is in LDS; step t - 1's tiles are free
      if (live) {
        if (t + P < n_steps) issue_step(t + P, s_nxt);
        s_nxt = slots_of(t + P + 1);
      }
      if constexpr (ROPE) {
        bool patch_any = false;   // some team patches the new row into its tile in this iteration (workgroup-uniform)
#pragma unroll
        for (int tt = 0; tt < TEAMS; ++tt) patch_any |= new_team[tt] && t == n_team[tt] - 1;
        if (patch_any) {
          if (has_new && t == n_steps - 1) {
            // the new token is the last key of the last step: its row goes into the tile over what was fetched
            // from the slot (every wave's loads of this step have landed: the barrier above)
            const int kl = (int)(len - 1 - k0) - 32 * t;           // 0 .. 31 inside the step
            char* row_tile = sm_team + (2 * (t % NSB) + (kl >> 4)) * MLR_TILE;
            const int r = kl & 15;
            if (tid < 64) {
              *reinterpret_cast<Vec<T, 8>*>(row_tile + MLR_ROWB * r + 16 * tid) = new_a;
            } else if (tid < 72 && !rp.is_neox) {
              *reinterpret_cast<Vec<T, 8>*>(row_tile + MLR_TAIL + 128 * r + 16 * ((tid - 64) ^ ((r >> 1) & 7))) = new_a;
            } else if (tid < 68) {
              *reinterpret_cast<Vec<T, 8>*>(row_tile + MLR_TAIL + 128 * r + 16 * ((tid - 64) ^ ((r >> 1) & 7))) = new_a;
              *reinterpret_cast<Vec<T, 8>*>(row_tile + MLR_TAIL + 128 * r + 16 * ((tid - 60) ^ ((r >> 1) & 7))) = new_b;
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
      }
      if (live) step_compute(t, sm_team + 2 * (t % NSB) * MLR_TILE);
    }
  }

  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if constexpr (TEAMS > 1) {
    // the teams' (reference point, sum, 512-feature accumulators) meet in LDS; team 0 carries the result out
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every team is done with its rings
    constexpr int TEAM_FLOATS = 4 * 8 * 64 * 4 + 4 * 64 * 2;
    float* mine = reinterpret_cast<float*>(sm_raw) + (team - 1) * TEAM_FLOATS;
    if (team > 0) {
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) reinterpret_cast<f32x4*>(mine)[(wave * 8 + dt) * 64 + lane] = acc[dt];
      reinterpret_cast<float2*>(mine + 4 * 8 * 64 * 4)[wave * 64 + lane] = float2{m, l};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (team == 0) {
#pragma unroll
      for (int tt = 1; tt < TEAMS; ++tt) {
        const float* other = reinterpret_cast<const float*>(sm_raw) + (tt - 1) * TEAM_FLOATS;
        const float2 ml = reinterpret_cast<const float2*>(other + 4 * 8 * 64 * 4)[wave * 64 + lane];
        const float mn = fmaxf(m, ml.x), msn = safe_m(mn);
        const float a0 = exp2f(m - msn), a1 = exp2f(ml.x - msn);
        l = l * a0 + ml.y * a1;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
          acc[dt] = acc[dt] * a0 + reinterpret_cast<const f32x4*>(other)[(wave * 8 + dt) * 64 + lane] * a1;
        m = mn;
      }
    }
  }
  const int s_out = p.num_splits / TEAMS;   // partials that leave the launch per (sequence, head)
  if (team == 0 && head0 + r16 < p.hq) {
    const int head = head0 + r16;
    const float inv = l > 0.f ? 1.f / l : 0.f;
    // lane (head r16, c4) holds features 128 wave + 16 dt + 4 c4 .. + 3
    if (s_out == 1) {
      T* dst = o + ((int64_t)b * p.hq + head) * MLA_RANK + 128 * wave + 4 * c4;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        Vec<T, 4> w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w.v[r] = from_f32<T>(acc[dt][r] * inv);
        stvec<T, 4>(dst + 16 * dt, w);
      }
    } else {
      // write-through, so that the workgroup that merges the splits (below) sees them without a fence
      // (the resources start at the workgroup's first head: wave-uniform; the lane adds its head and features)
      const int64_t ph0 = ((int64_t)b * p.hq + head0) * s_out;
      const int cell = r16 * s_out + (int)blockIdx.y;
      const auto rs = wt_rsrc(p.part_o + ph0 * MLA_RANK, 16 * s_out * MLA_RANK * 4);
#pragma unroll
      for (int dt = 0; dt < 8; ++dt)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[dt] * inv), rs,
                                               (cell * MLA_RANK + 128 * wave + 4 * c4 + 16 * dt) * 4, 0, 16);
      if (c4 == 0 && wave == 0)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, l > 0.f ? m * kLn2 + logf(l) : -INFINITY),
                                              wt_rsrc(p.part_lse + ph0, 16 * s_out * 4), cell * 4, 0, 16);
    }
  }
  if (s_out > 1 && p.tickets) {
    const int nheads = p.hq - head0 < 16 ? p.hq - head0 : 16;
    DecodeParams po = p;
    po.num_splits = s_out;
    merge_splits_if_last<T, 4>(po, o, b, head0, nheads, MLA_RANK, b * (int)gridDim.x + (int)blockIdx.x, reinterpret_cast<float*>(sm_raw));
  }
}

// ------------------------------------------------------------------------------------------
// MLA decode over an fp8 (e4m3fn) latent cache through an LDS ring: rows of 576 BYTES land in LDS as they
// are (global_load_lds_dwordx4) and are widened to T exactly when a fragment is used (one
// v_cvt_scalef32_pk_*_fp8 per two values, scale 1: every e4m3 value is a value of T).
// A step is 32 keys.  LDS image of a step: row r at 592 r (+ 128 for r >= 16): 576 bytes + 16 of padding, so
// the score reads (ds_read_b64, 16 rows x 8 bytes) and the transposed value reads (ds_read_b64_tr_b8, rows
// r .. r + 3 and r + 16 .. r + 19 per 16 lanes) are conflict-free.  Wave w fetches rows 8 w .. 8 w + 7 of a
// step as five wave-wide loads over the 296 sixteen-byte pieces of its 4736-byte region (a lane's piece never
// crosses a row; the 37th piece of a row is padding and its lane stays idle).
// The scores are NOT computed four times: wave w takes tile w >> 1 (16 keys) and half w & 1 of the 18 feature
// steps, the four partial tiles meet in LDS (one more barrier per step) and every wave adds the two halves
// of both tiles in the same order -- identical softmax statistics in all waves, as in the 16-bit ring kernel,
// at a quarter of its score work and LDS traffic, which is what the conversion instructions need to fit.
// P.V: 16x16x32 steps whose 32 key slots are {4 c + j, 16 + 4 c + j} for lane group c -- the order the score
// accumulators hold them in -- with the value fragment read transposed from the rows of exactly those keys.
// grid (ceil(hq / 16), num_splits, B), block 256; NSB step buffers (NSB - 1 steps in flight).
// ------------------------------------------------------------------------------------------
constexpr int M8_ROWB = 592;                          // bytes between rows of a step image
constexpr int M8_SHIFT = 128;                         // rows 16 .. 31 sit 32 banks further
constexpr int M8_STEP = 32 * M8_ROWB + M8_SHIFT;      // 19072
constexpr int M8_XCH = 2 * 4 * 64 * 16;               // partial scores: two parities x four waves x 1 KiB
constexpr int m8_lds_bytes(int nsb) { return nsb * M8_STEP + M8_XCH; }
__device__ __forceinline__ constexpr int m8_row(int r) { return r * M8_ROWB + (r >= 16 ? M8_SHIFT : 0); }

template <typename T> struct Fp8Widen;
template <> struct Fp8Widen<bf16_t> {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ v2 run(int w, bool hi) {
    return hi ? __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true) : __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false);
  }
};
template <> struct Fp8Widen<f16_t> {
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ v2 run(int w, bool hi) {
    return hi ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true) : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false);
  }
};
// eight e4m3fn bytes -> one MFMA fragment of eight T (exact)
template <typename T> __device__ __forceinline__ typename Mfma<T>::frag widen8(uint2 raw) {
  typedef typename Fp8Widen<T>::v2 v2;
  struct alignas(16) Quad { v2 a, b, c, d; };
  const Quad q{Fp8Widen<T>::run((int)raw.x, false), Fp8Widen<T>::run((int)raw.x, true),
               Fp8Widen<T>::run((int)raw.y, false), Fp8Widen<T>::run((int)raw.y, true)};
  return __builtin_bit_cast(typename Mfma<T>::frag, q);
}

// ROPE: as in the 16-bit ring kernel -- the query's rope part is rotated in registers (the waves that own feature
// steps 16 and 17), the new token's row [k_nope | rope(k_pe)] is built, quantised with the scatter kernel's
// arithmetic, written to the cache by one workgroup per sequence and patched into the LDS image by every
// workgroup whose key range ends at it.
// TEAMS = 2: a workgroup of 512 threads is two teams of four waves, each with its own ring, working on two
// consecutive KV splits (2 y, 2 y + 1) of the launch's split plan; the teams' results meet in LDS at the end and ONE
// partial leaves the workgroup.  Same waves per CU and bytes in flight as two 256-thread workgroups, half the
// workgroups to dispatch, half the partials to write and to merge (the merge of S partials of 16 heads x 512
// features is what a launch pays for its splits: 64 x 16 heads, 64 keys: 9 / 15 / 18 / 26 us at 1 / 2 / 4 / 8).
// grid.y = num_splits / TEAMS; barriers are workgroup-wide, so both teams walk max(steps) iterations.
template <typename T, bool CAP, int NSB, bool ROPE = false, int TEAMS = 1>
__global__ __launch_bounds__(256 * TEAMS) void decode_attn_mla_ring8_kernel(
    const T* __restrict__ q, const uint8_t* __restrict__ kvbuf, T* __restrict__ o, DecodeParams p, MlaRope rp) {
  constexpr int NH = MLA_D / 64;   // 9 of the 18 score k-steps per wave
  using frag = typename Mfma<T>::frag;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  constexpr int P = NSB - 1;       // steps requested ahead of the one being computed
  static_assert(P >= 1 && P <= 3, "the wait below knows 0, 5, 10 or 15 younger loads");
  static_assert(MLA_D == 576 && MLA_RANK == 512, "the step image is laid out for 512 + 64 byte rows");
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];   // TEAMS * m8_lds_bytes(NSB)

  const int head0 = blockIdx.x * 16, b = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int team = TEAMS > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;
  const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);   // within the team
  const int tid = threadIdx.x & 255;                                         // within the team
  const int split = blockIdx.y * TEAMS + team;
  char* sm_team = sm_raw + team * m8_lds_bytes(NSB);
  const int r16 = lane & 15, c4 = lane >> 4;
  const int tile = wave >> 1, half = wave & 1;

  int64_t page_off, len;
  seq_range(p, b, page_off, len);
  const int64_t chunk = split_chunk(p, len);
  const int64_t k0 = (int64_t)split * chunk;
  const int64_t k1 = k0 + chunk < len ? k0 + chunk : len;
  const int n_steps = k1 > k0 ? (int)((k1 - k0 + 31) / 32) : 0;
  const float qmul = CAP ? p.sm_scale : p.sm_scale * kLog2e;
  // every team's step count and whether its range ends at the new token: the barriers are shared
  int n_team[TEAMS], n_all = 0;
  bool new_team[TEAMS];
#pragma unroll
  for (int tt = 0; tt < TEAMS; ++tt) {
    const int64_t a0 = (int64_t)(blockIdx.y * TEAMS + tt) * chunk;
    const int64_t a1 = a0 + chunk < len ? a0 + chunk : len;
    n_team[tt] = a1 > a0 ? (int)((a1 - a0 + 31) / 32) : 0;
    new_team[tt] = ROPE && a1 == len && len > a0;
    n_all = n_team[tt] > n_all ? n_team[tt] : n_all;
  }

  // ROPE: threads 0..63 hold eight bytes of the new row's latent part, threads 64..71 (interleaved pairs) or
  // 64..67 (neox pairs: a piece and its partner 32 features on) of its rotated rope part.
  const bool has_new = ROPE && k1 == len && len > k0;   // this workgroup's range ends at the new token
  uint2 new_qa = uint2{0u, 0u}, new_qb = uint2{0u, 0u};
  int64_t rope_pos = 0;
  if constexpr (ROPE) {
    rope_pos = rp.positions[b];
    if (has_new) {
      Vec<T, 8> new_a, new_b;
#pragma unroll
      for (int j = 0; j < 8; ++j) new_a.v[j] = new_b.v[j] = from_f32<T>(0.f);
      if (tid < 64) {
        new_a = ldvec<T, 8>(static_cast<const T*>(rp.k_nope) + (int64_t)b * rp.k_nope_stride + 8 * tid);
      } else if (tid < 72) {
        const T* kp = static_cast<const T*>(rp.k_pe) + (int64_t)b * rp.k_pe_stride;
        const int c = tid - 64;
        if (!rp.is_neox) {
          const Vec<T, 8> x = ldvec<T, 8>(kp + 8 * c);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = 4 * c + j;
            rope_pair<T>(to_f32(x.v[2 * j]), to_f32(x.v[2 * j + 1]), mla_cs_t<T>(rp, rope_pos, i),
                         mla_cs_t<T>(rp, rope_pos, MLA_ROPE / 2 + i), new_a.v[2 * j], new_a.v[2 * j + 1]);
          }
        } else if (c < 4) {
          const Vec<T, 8> x1 = ldvec<T, 8>(kp + 8 * c), x2 = ldvec<T, 8>(kp + MLA_ROPE / 2 + 8 * c);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int i = 8 * c + j;
            rope_pair<T>(to_f32(x1.v[j]), to_f32(x2.v[j]), mla_cs_t<T>(rp, rope_pos, i),
                         mla_cs_t<T>(rp, rope_pos, MLA_ROPE / 2 + i), new_a.v[j], new_b.v[j]);
          }
        }
      }
      new_qa = quant8_fp8<T>(new_a, 1.0f);   // the scatter kernel's rounding (clamp to +-448, nearest even)
      new_qb = quant8_fp8<T>(new_b, 1.0f);
      if (blockIdx.x == 0) {   // one workgroup per sequence writes the row to the cache
        uint8_t* dst = const_cast<uint8_t*>(kvbuf) + rp.out_loc[b] * MLA_D;
        if (tid < 64) *reinterpret_cast<uint2*>(dst + 8 * tid) = new_qa;
        else if (tid < 72 && !rp.is_neox) *reinterpret_cast<uint2*>(dst + MLA_RANK + 8 * (tid - 64)) = new_qa;
        else if (tid < 68) {
          *reinterpret_cast<uint2*>(dst + MLA_RANK + 8 * (tid - 64)) = new_qa;
          *reinterpret_cast<uint2*>(dst + MLA_RANK + MLA_ROPE / 2 + 8 * (tid - 64)) = new_qb;
        }
      }
    }
  }

  // page-table reads through the scalar unit (wave-uniform index, read-only data)
  const int32_t __attribute__((address_space(4)))* pt =
      (const int32_t __attribute__((address_space(4)))*)(p.page_table + page_off);
  const int64_t last_key = len - 1;
  struct Slots { int32_t s[8]; };   // rows 8 wave .. + 7 of a step
  auto slots_of = [&](int step) -> Slots {
    const int64_t key = k0 + (int64_t)step * 32 + 8 * wave;
    Slots sl;
#pragma unroll
    for (int j = 0; j < 8; ++j) sl.s[j] = pt[key + j < last_key ? key + j : last_key];
    return sl;
  };
  const unsigned ring = lds_byte_address(sm_team);
  // load i (0 .. 4) of a step: this lane's piece 64 i + lane of the wave's region = (row, 16-byte chunk)
  int pj[5], poff[5];
  bool pact[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int g = 64 * i + lane;
    const int j = g / 37, c = g - 37 * j;
    pact[i] = g < 296 && c < 36;
    pj[i] = j - (64 * i) / 37;   // 0 .. 2: which of the (at most three) rows this load touches
    poff[i] = 16 * (c < 36 ? c : 0);
  }
  auto issue_step = [&](int step, const Slots& sl) {
    const unsigned base = ring + (unsigned)(step % NSB) * M8_STEP + (unsigned)m8_row(8 * wave);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      constexpr int JB[5] = {0, 1, 3, 5, 6};
      const int jb = JB[i];
      const int32_t sa = sl.s[jb], sb = sl.s[jb + 1 < 8 ? jb + 1 : 7], sc = sl.s[jb + 2 < 8 ? jb + 2 : 7];
      const uint32_t slot = (uint32_t)(pj[i] == 0 ? sa : (pj[i] == 1 ? sb : sc));
      if (pact[i])
        load_to_lds_16_v<true>(kvbuf + (uint64_t)slot * MLA_D + poff[i], base + 1024u * i);
    }
  };

  float m = -INFINITY, l = 0.f;
  f32x4 acc[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // read addresses inside a step image (bytes)
  const int k_rd = m8_row(16 * tile + r16) + 32 * NH * half + 8 * c4;   // + 32 s'
  // value fragment of lane (group c4, column r16): lane 2 q + pp supplies row q, columns 8 pp .. + 7 of its group's
  // eight keys {4 c4 + (q & 3) + 16 (q >> 2)}
  const int vq = r16 >> 1;
  const int v_rd = m8_row(4 * c4 + (vq & 3) + 16 * (vq >> 2)) + 128 * wave + 8 * (r16 & 1);   // + 16 dt
  char* xch = sm_team + NSB * M8_STEP;

  Slots s_nxt{};
  if (n_steps > 0) {   // (an empty sequence has no page-table entry to read)
#pragma unroll
    for (int i = 0; i < P; ++i) {
      s_nxt = slots_of(i);   // clamped to the last key when past the end
      if (i < n_steps) issue_step(i, s_nxt);
    }
    s_nxt = slots_of(P);
  }

  // B operand of this wave's score steps: head r16, features 32 (9 half + s') + 8 c4 .. + 7 (fetched after the
  // first tiles were requested and waited for before the loop, as in the 16-bit kernel)
  frag qf[NH];
  {
    const int hq_c = head0 + r16 < p.hq ? head0 + r16 : p.hq - 1;   // rows of heads past hq are computed and dropped
    const T* qp = q + (int64_t)b * p.q_stride + (int64_t)hq_c * MLA_D + 32 * NH * half + 8 * c4;
#pragma unroll
    for (int s = 0; s < NH; ++s) qf[s] = *reinterpret_cast<const frag*>(qp + 32 * s);
    if constexpr (ROPE) {
      if (half == 1) {
        // rope part = feature steps 16 and 17 = this wave's steps NH - 2 and NH - 1: the lane holds features
        // 8 c4 .. + 7 and 32 + 8 c4 .. + 7 of its head
        Vec<T, 8> a = __builtin_bit_cast(Vec<T, 8>, qf[NH - 2]), b2 = __builtin_bit_cast(Vec<T, 8>, qf[NH - 1]);
        if (!rp.is_neox) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ia = 4 * c4 + j, ib = 16 + 4 * c4 + j;
            rope_pair<T>(to_f32(a.v[2 * j]), to_f32(a.v[2 * j + 1]), mla_cs_t<T>(rp, rope_pos, ia),
                         mla_cs_t<T>(rp, rope_pos, MLA_ROPE / 2 + ia), a.v[2 * j], a.v[2 * j + 1]);
            rope_pair<T>(to_f32(b2.v[2 * j]), to_f32(b2.v[2 * j + 1]), mla_cs_t<T>(rp, rope_pos, ib),
                         mla_cs_t<T>(rp, rope_pos, MLA_ROPE / 2 + ib), b2.v[2 * j], b2.v[2 * j + 1]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int i = 8 * c4 + j;
            rope_pair<T>(to_f32(a.v[j]), to_f32(b2.v[j]), mla_cs_t<T>(rp, rope_pos, i),
                         mla_cs_t<T>(rp, rope_pos, MLA_ROPE / 2 + i), a.v[j], b2.v[j]);
          }
        }
        qf[NH - 2] = __builtin_bit_cast(frag, a);
        qf[NH - 1] = __builtin_bit_cast(frag, b2);
      }
    }
#pragma unroll
    for (int s = 0; s < NH; ++s) asm volatile("" ::"v"(qf[s]));
  }

  for (int t = 0; t < n_all; ++t) {
    const bool live = t < n_steps;   // (team-uniform; a team that has run out of steps keeps the barriers company)
    if (live) {
      // this wave's part of step t has landed when at most the loads (5 per step) of the younger steps are out
      const int younger = min(P - 1, n_steps - 1 - t);
      if (P >= 3 && younger >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (P >= 2 && younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // and its reads of step t - 1 have returned
    __builtin_amdgcn_s_barrier();                         // every wave's part of step t is in LDS; step t - 1's image is free
    if (live) {
      if (t + P < n_steps) issue_step(t + P, s_nxt);
      s_nxt = slots_of(t + P + 1);
    }
    char* img = sm_team + (t % NSB) * M8_STEP;
    if constexpr (ROPE) {
      bool patch_any = false;   // some team patches the new row into its image in this iteration (workgroup-uniform)
#pragma unroll
      for (int tt = 0; tt < TEAMS; ++tt) patch_any |= new_team[tt] && t == n_team[tt] - 1;
      if (patch_any) {
       if (has_new && t == n_steps - 1) {
        // the new token is the last key of the last step: its bytes go into the image over what was fetched from
        // the slot (every wave's loads of this step have landed: the barrier above)
        const int kl = (int)(len - 1 - k0) - 32 * t;           // 0 .. 31 inside the step
        char* row = img + m8_row(kl);
        if (tid < 64) *reinterpret_cast<uint2*>(row + 8 * tid) = new_qa;
        else if (tid < 72 && !rp.is_neox) *reinterpret_cast<uint2*>(row + MLA_RANK + 8 * (tid - 64)) = new_qa;
        else if (tid < 68) {
          *reinterpret_cast<uint2*>(row + MLA_RANK + 8 * (tid - 64)) = new_qa;
          *reinterpret_cast<uint2*>(row + MLA_RANK + MLA_ROPE / 2 + 8 * (tid - 64)) = new_qb;
        }
       }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }

    // this wave's partial scores: 16 keys of its tile x 16 heads over its nine feature steps
    // (ds_read_b64 by hand: the compiler pairs these reads into ds_read2_b64, which banks modulo 32 over 16-lane
    // groups -- rows r and r + 8 of the 592-byte image then collide, 32 conflict cycles per wave and step)
    f32x4* xw = reinterpret_cast<f32x4*>(xch + (t & 1) * 4096);
    if (live) {
    unsigned long long kraw[NH];
    const unsigned k_addr = lds_byte_address(img + k_rd);
#pragma unroll
    for (int s = 0; s < NH; ++s) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(kraw[s]) : "v"(k_addr), "n"(32 * s));
    static_assert(NH == 9, "the wait below names nine fragments");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(kraw[0]), "+v"(kraw[1]), "+v"(kraw[2]), "+v"(kraw[3]), "+v"(kraw[4]), "+v"(kraw[5]), "+v"(kraw[6]),
                   "+v"(kraw[7]), "+v"(kraw[8]));
    f32x4 part = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NH; ++s)
      part = Mfma<T>::run(widen8<T>(uint2{(unsigned)kraw[s], (unsigned)(kraw[s] >> 32)}), qf[s], part);
    xw[wave * 64 + lane] = part;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!live) continue;
    const f32x4 sa = xw[lane] + xw[64 + lane], sb = xw[128 + lane] + xw[192 + lane];

    const int64_t kb0 = k0 + (int64_t)t * 32 + 4 * c4;
    float x[8], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float v = (r < 4 ? sa[r & 3] : sb[r & 3]) * qmul;
      if (CAP) v = p.logit_cap * tanhf(v / p.logit_cap) * kLog2e;
      x[r] = kb0 + (r & 3) + 16 * (r >> 2) < k1 ? v : -INFINITY;
      mx = fmaxf(mx, x[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // lazy rescale: the reference point moves only when a step's maximum exceeds it by 2^8
    const bool need = mx > m + 8.f;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
      const float mnew = need ? mx : m;
      const float alpha = exp2f(m - safe_m(mnew));
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) acc[dt] *= alpha;
      m = mnew;
    }
    const float ms = safe_m(m);
    float psum = 0.f;
    Vec<T, 8> p8;   // key slots 8 c4 + j of the P.V step: keys 4 c4 + j (j < 4), 16 + 4 c4 + j - 4 (j >= 4)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float pe = exp2f(x[r] - ms);
      psum += pe;
      p8.v[r] = from_f32<T>(pe);
    }
    l += psum;
    const frag pf = __builtin_bit_cast(frag, p8);
    i32x2 vraw[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
      vraw[dt] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((i32x2 __attribute__((address_space(3)))*)(img + v_rd + 16 * dt));
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
      acc[dt] = Mfma<T>::run(widen8<T>(uint2{(unsigned)vraw[dt][0], (unsigned)vraw[dt][1]}), pf, acc[dt]);
  }

  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if constexpr (TEAMS > 1) {
    // the teams' (reference point, sum, 512-feature accumulators) meet in LDS; team 0 carries the result out
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every team is done with its rings
    constexpr int TEAM_FLOATS = 4 * 8 * 64 * 4 + 4 * 64 * 2;
    float* mine = reinterpret_cast<float*>(sm_raw) + (team - 1) * TEAM_FLOATS;
    if (team > 0) {
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) reinterpret_cast<f32x4*>(mine)[(wave * 8 + dt) * 64 + lane] = acc[dt];
      reinterpret_cast<float2*>(mine + 4 * 8 * 64 * 4)[wave * 64 + lane] = float2{m, l};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (team == 0) {
#pragma unroll
      for (int tt = 1; tt < TEAMS; ++tt) {
        const float* other = reinterpret_cast<const float*>(sm_raw) + (tt - 1) * TEAM_FLOATS;
        const float2 ml = reinterpret_cast<const float2*>(other + 4 * 8 * 64 * 4)[wave * 64 + lane];
        const float mn = fmaxf(m, ml.x), msn = safe_m(mn);
        const float a0 = exp2f(m - msn), a1 = exp2f(ml.x - msn);
        l = l * a0 + ml.y * a1;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
          acc[dt] = acc[dt] * a0 + reinterpret_cast<const f32x4*>(other)[(wave * 8 + dt) * 64 + lane] * a1;
        m = mn;
      }
    }
  }
  const int s_out = p.num_splits / TEAMS;   // partials that leave the launch per (sequence, head)
  if (team == 0 && head0 + r16 < p.hq) {
    const int head = head0 + r16;
    const float inv = l > 0.f ? 1.f / l : 0.f;
    // lane (head r16, c4) holds features 128 wave + 16 dt + 4 c4 .. + 3
    if (s_out == 1) {
      T* dst = o + ((int64_t)b * p.hq + head) * MLA_RANK + 128 * wave + 4 * c4;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        Vec<T, 4> w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w.v[r] = from_f32<T>(acc[dt][r] * inv);
        stvec<T, 4>(dst + 16 * dt, w);
      }
    } else {
      const int64_t ph0 = ((int64_t)b * p.hq + head0) * s_out;
      const int cell = r16 * s_out + (int)blockIdx.y;
      const auto rs = wt_rsrc(p.part_o + ph0 * MLA_RANK, 16 * s_out * MLA_RANK * 4);
#pragma unroll
      for (int dt = 0; dt < 8; ++dt)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[dt] * inv), rs,
                                               (cell * MLA_RANK + 128 * wave + 4 * c4 + 16 * dt) * 4, 0, 16);
      if (c4 == 0 && wave == 0)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, l > 0.f ? m * kLn2 + logf(l) : -INFINITY),
                                              wt_rsrc(p.part_lse + ph0, 16 * s_out * 4), cell * 4, 0, 16);
    }
  }
  if (s_out > 1 && p.tickets) {
    const int nheads = p.hq - head0 < 16 ? p.hq - head0 : 16;
    DecodeParams po = p;
    po.num_splits = s_out;
    merge_splits_if_last<T, 4>(po, o, b, head0, nheads, MLA_RANK, b * (int)gridDim.x + (int)blockIdx.x, reinterpret_cast<float*>(sm_raw));
  }
}

// ------------------------------------------------------------------------------------------
// MLA extend in the absorbed form over cached rows: the new tokens of a request attend to the request's
// cached prefix and, causally, to each other (their latent rows are in the cache already).  The ring of the
// decode kernel above, shared by FOUR consecutive new tokens of one request: wave w owns token 4 ty + w,
// computes that token's 16 heads x 32 keys of scores per step (nothing is computed twice) and accumulates all
// 512 value features for it (128 accumulator registers), so every latent row fetched into LDS serves 64
// query rows instead of 16 and the score MFMAs are not repeated per wave.  Rows are requested up to the
// LAST token's length; a wave masks keys at or beyond its own token's length and skips steps wholly beyond it.
// A deep ring (4 step buffers, 148 KiB) and up to 512 registers per lane: one workgroup per CU.
// grid (ceil(hq / 16), ceil(max new tokens / 4), B), block 256; qo_indptr[b] = first row of request b in q / o.
// ------------------------------------------------------------------------------------------
constexpr int MLX_TOKENS = 4;
template <typename T, bool CAP>
__global__ __launch_bounds__(256) void extend_attn_mla_tile_kernel(
    const T* __restrict__ q, const T* __restrict__ kvbuf, T* __restrict__ o, DecodeParams p,
    const int32_t* __restrict__ qo_indptr) {
  constexpr int NS = MLA_D / 32;   // 18 score k-steps
  constexpr int NSB = 4, P = NSB - 1;
  using frag = typename Mfma<T>::frag;
  using frag4 = typename Mfma16<T>::frag;
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];   // mlr_lds_bytes(NSB)

  // the last tokens of a request see the most keys: their tiles are dispatched first
  const int head0 = blockIdx.x * 16, ty = gridDim.y - 1 - blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, c4 = lane >> 4;

  const int q0 = qo_indptr[b], n_new = qo_indptr[b + 1] - q0;
  if (ty * MLX_TOKENS >= n_new) return;
  const int64_t page_off = p.req_pool_indices[b] * p.table_stride;
  const int64_t prefix = p.seq_lens[b] - n_new;
  const bool valid = ty * MLX_TOKENS + wave < n_new;                      // wave-uniform
  const int tok = valid ? ty * MLX_TOKENS + wave : n_new - 1;            // a spare wave repeats the last token and stores nothing
  const int64_t len_w = prefix + tok + 1;                                 // keys this wave's token sees
  const int last_tok = min(ty * MLX_TOKENS + MLX_TOKENS, n_new) - 1;
  const int64_t len = prefix + last_tok + 1;                              // keys the workgroup fetches
  const int n_steps = (int)((len + 31) / 32);
  const float qmul = CAP ? p.sm_scale : p.sm_scale * kLog2e;

  const int32_t __attribute__((address_space(4)))* pt =
      (const int32_t __attribute__((address_space(4)))*)(p.page_table + page_off);
  const int64_t last_key = len - 1;
  struct Slots { int32_t a0, a1, a2, a3, b0, b1, b2, b3; };
  auto slot_of = [&](int64_t key) -> int32_t { return pt[key < last_key ? key : last_key]; };
  auto slots_of = [&](int step) -> Slots {
    const int64_t key = (int64_t)step * 32 + 4 * wave;
    return Slots{slot_of(key), slot_of(key + 1), slot_of(key + 2), slot_of(key + 3),
                 slot_of(key + 16), slot_of(key + 17), slot_of(key + 18), slot_of(key + 19)};
  };
  const unsigned ring = lds_byte_address(sm_raw);
  const uint64_t kv_base = (uint64_t)(uintptr_t)kvbuf;
  const uint64_t row_bytes = (uint64_t)p.k_slot_stride * sizeof(T);
  const unsigned lane16 = 16u * lane;
  const int trow = lane >> 3, tch = lane & 7;
  const unsigned tsw = MLA_RANK * sizeof(T) + 16 * (tch ^ (((4 * wave + trow) >> 1) & 7));
  const int tm0 = -(int)(trow == 0), tm1 = -(int)(trow == 1), tm2 = -(int)(trow == 2), tm3 = -(int)(trow == 3);
  auto issue_tile = [&](int buf, int32_t s0, int32_t s1, int32_t s2, int32_t s3) {
    const unsigned base = ring + (unsigned)buf * MLR_TILE + (unsigned)(4 * wave) * MLR_ROWB;
    load_to_lds_16_s(kv_base + (uint64_t)(uint32_t)s0 * row_bytes, lane16, base);
    load_to_lds_16_s(kv_base + (uint64_t)(uint32_t)s1 * row_bytes, lane16, base + MLR_ROWB);
    load_to_lds_16_s(kv_base + (uint64_t)(uint32_t)s2 * row_bytes, lane16, base + 2 * MLR_ROWB);
    load_to_lds_16_s(kv_base + (uint64_t)(uint32_t)s3 * row_bytes, lane16, base + 3 * MLR_ROWB);
    const uint32_t ts = (uint32_t)((s0 & tm0) | (s1 & tm1) | (s2 & tm2) | (s3 & tm3));
    if (lane < 32)
      load_to_lds_16_v(reinterpret_cast<const char*>(kvbuf) + (uint64_t)ts * row_bytes + tsw,
                       ring + (unsigned)buf * MLR_TILE + MLR_TAIL + (unsigned)(4 * wave) * 128);
  };
  auto issue_step = [&](int step, const Slots& sl) {
    const int buf = 2 * (step % NSB);
    issue_tile(buf, sl.a0, sl.a1, sl.a2, sl.a3);
    issue_tile(buf + 1, sl.b0, sl.b1, sl.b2, sl.b3);
  };

  float m = -INFINITY, l = 0.f;
  f32x4 acc[32];
#pragma unroll
  for (int dt = 0; dt < 32; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int k_rd = MLR_ROWB * r16 + 16 * c4;
  const int kt_rd0 = MLR_TAIL + 128 * r16 + 16 * (c4 ^ ((r16 >> 1) & 7));
  const int kt_rd1 = MLR_TAIL + 128 * r16 + 16 * ((4 + c4) ^ ((r16 >> 1) & 7));
  const int v_rd = MLR_ROWB * (4 * c4 + (r16 >> 2)) + 8 * (r16 & 3);   // + 32 (feature block of 16)

  Slots s_nxt{};
#pragma unroll
  for (int i = 0; i < P; ++i) {
    s_nxt = slots_of(i);
    if (i < n_steps) issue_step(i, s_nxt);
  }
  s_nxt = slots_of(P);

  frag qf[NS];
  {
    const int hq_c = head0 + r16 < p.hq ? head0 + r16 : p.hq - 1;
    const T* qp = q + (int64_t)(q0 + tok) * p.q_stride + (int64_t)hq_c * MLA_D + 8 * c4;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const frag*>(qp + 32 * s);
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" ::"v"(qf[s]));
  }

  auto step_compute = [&](int t, const char* tile) {
    f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int AHEAD = 4;
    auto k_off = [&](int s) { return s < 16 ? k_rd + 64 * s : (s == 16 ? kt_rd0 : kt_rd1); };
    frag ka[AHEAD], kb[AHEAD];
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) {
      ka[s] = *reinterpret_cast<const frag*>(tile + k_off(s));
      kb[s] = *reinterpret_cast<const frag*>(tile + MLR_TILE + k_off(s));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      sa = Mfma<T>::run(ka[s % AHEAD], qf[s], sa);
      sb = Mfma<T>::run(kb[s % AHEAD], qf[s], sb);
      __builtin_amdgcn_sched_barrier(0);
      if (s + AHEAD < NS) {
        ka[s % AHEAD] = *reinterpret_cast<const frag*>(tile + k_off(s + AHEAD));
        kb[s % AHEAD] = *reinterpret_cast<const frag*>(tile + MLR_TILE + k_off(s + AHEAD));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int64_t kb0 = (int64_t)t * 32 + 4 * c4;
    float x[8], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float v = (r < 4 ? sa[r & 3] : sb[r & 3]) * qmul;
      if (CAP) v = p.logit_cap * tanhf(v / p.logit_cap) * kLog2e;
      x[r] = kb0 + (r & 3) + 16 * (r >> 2) < len_w ? v : -INFINITY;
      mx = fmaxf(mx, x[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const bool need = mx > m + 8.f;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
      const float mnew = need ? mx : m;
      const float alpha = exp2f(m - safe_m(mnew));
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 32; ++dt) acc[dt] *= alpha;
      m = mnew;
    }
    const float ms = safe_m(m);
    float psum = 0.f;
    T pa4[4], pb4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pa = exp2f(x[r] - ms), pb = exp2f(x[4 + r] - ms);
      psum += pa + pb;
      pa4[r] = from_f32<T>(pa);
      pb4[r] = from_f32<T>(pb);
    }
    l += psum;
    const frag4 pa = __builtin_bit_cast(frag4, pa4), pb = __builtin_bit_cast(frag4, pb4);
#pragma unroll
    for (int h = 0; h < 8; ++h) {   // 8 batches of 4 feature blocks
      typename TrRead<T>::v4 va[4], vb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va[i] = TrRead<T>::run(reinterpret_cast<const T*>(tile + v_rd + 32 * (4 * h + i)));
        vb[i] = TrRead<T>::run(reinterpret_cast<const T*>(tile + MLR_TILE + v_rd + 32 * (4 * h + i)));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[4 * h + i] = Mfma16<T>::run(__builtin_bit_cast(frag4, va[i]), pa, acc[4 * h + i]);
        acc[4 * h + i] = Mfma16<T>::run(__builtin_bit_cast(frag4, vb[i]), pb, acc[4 * h + i]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int t = 0; t < n_steps; ++t) {
    const int younger = min(P - 1, n_steps - 1 - t);
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + P < n_steps) issue_step(t + P, s_nxt);
    s_nxt = slots_of(t + P + 1);
    if ((int64_t)t * 32 < len_w) step_compute(t, sm_raw + 2 * (t % NSB) * MLR_TILE);
  }

  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (valid && head0 + r16 < p.hq) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    // lane (head r16, c4) holds features 16 dt + 4 c4 .. + 3
    T* dst = o + ((int64_t)(q0 + tok) * p.hq + head0 + r16) * MLA_RANK + 4 * c4;
#pragma unroll
    for (int dt = 0; dt < 32; ++dt) {
      Vec<T, 4> w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w.v[r] = from_f32<T>(acc[dt][r] * inv);
      stvec<T, 4>(dst + 16 * dt, w);
    }
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
  const float tot = merge_weights(lse, num_splits, nullptr);
  float mx = -INFINITY;
  for (int s = 0; s < num_splits; ++s) mx = fmaxf(mx, lse[s]);
  const float ms = safe_m(mx);
  for (int d = threadIdx.x; d < Dv; d += 128) {
    float a = 0.f;
    for (int s = 0; s < num_splits; ++s)
      a = merge_step(expf(lse[s] - ms), part_o[(bh * num_splits + s) * Dv + d], a);
    o[bh * Dv + d] = from_f32<T>(merge_finish(a, tot));
  }
}

// 0 = P.V on MFMA, 1 = P.V on the vector ALU: an ablation switch of the development build
// (mi355_debug_set key 6); a compile-time zero in the product library.
#ifdef MI355_DEV_BUILD
int g_decode_variant = 0;
#else
static constexpr int g_decode_variant = 0;
#endif

template <typename T, int D, int U, bool CAP>
static void launch_decode_dc(const T* q, const T* kb, const T* vb, T* o, const DecodeParams& p,
                             int64_t bs, bool kv8, hipStream_t st) {   // kv8: the KV cache holds e4m3fn bytes
  const int gtotal = p.hq / p.hkv;
  const int G = gtotal % 8 == 0 ? 8 : gtotal % 4 == 0 ? 4 : gtotal % 2 == 0 ? 2 : 1;
  const dim3 grid((unsigned)(p.hkv * (gtotal / G)), (unsigned)p.num_splits, (unsigned)bs);
  const dim3 block(256);
  if constexpr (D == 128 && !std::is_same<T, float>::value) {
    if (kv8) {
      switch (G) {
        case 8: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 8, CAP, false, true, true>), grid, block, 0, st, q, kb, vb, o, p, FusedQkv()); break;
        case 4: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 4, CAP, false, true, true>), grid, block, 0, st, q, kb, vb, o, p, FusedQkv()); break;
        case 2: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 2, CAP, false, true, true>), grid, block, 0, st, q, kb, vb, o, p, FusedQkv()); break;
        default: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 1, CAP, false, true, true>), grid, block, 0, st, q, kb, vb, o, p, FusedQkv()); break;
      }
      return;
    }
    if (g_decode_variant == 0) {
      switch (G) {
        case 8: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 8, CAP, false, false, true>), grid, block, 0, st, q, kb, vb, o, p, FusedQkv()); break;
        case 4: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 4, CAP, false, false, true>), grid, block, 0, st, q, kb, vb, o, p, FusedQkv()); break;
        case 2: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 2, CAP, false, false, true>), grid, block, 0, st, q, kb, vb, o, p, FusedQkv()); break;
        default: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 1, CAP, false, false, true>), grid, block, 0, st, q, kb, vb, o, p, FusedQkv()); break;
      }
      return;
    }
#ifdef MI355_DEV_BUILD
    if (g_decode_variant == 3) {   // ablation: K rows loaded straight into MFMA fragments (16 rows x 64 B per instruction)
      switch (G) {
        case 8: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 8, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
        case 4: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 4, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
        case 2: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 2, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
        default: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 1, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
      }
      return;
    }
    // ablation: P.V on the vector ALU (variant 1)
    switch (G) {
      case 8: hipLaunchKernelGGL((decode_attn_mfma_kernel<T, 8, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
      case 4: hipLaunchKernelGGL((decode_attn_mfma_kernel<T, 4, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
      case 2: hipLaunchKernelGGL((decode_attn_mfma_kernel<T, 2, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
      default: hipLaunchKernelGGL((decode_attn_mfma_kernel<T, 1, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
    }
#endif
    return;
  }
  switch (G) {
    case 8: hipLaunchKernelGGL((decode_attn_kernel<T, D, 8, (U > 2 ? 2 : U), CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
    case 4: hipLaunchKernelGGL((decode_attn_kernel<T, D, 4, U, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
    case 2: hipLaunchKernelGGL((decode_attn_kernel<T, D, 2, U, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
    default: hipLaunchKernelGGL((decode_attn_kernel<T, D, 1, U, CAP>), grid, block, 0, st, q, kb, vb, o, p); break;
  }
}

template <typename T, int D, int U>
static bool launch_decode_d(const T* q, const T* kb, const T* vb, T* o, const DecodeParams& p,
                            int64_t bs, bool kv8, hipStream_t st) {
  if (p.logit_cap > 0.f) launch_decode_dc<T, D, U, true>(q, kb, vb, o, p, bs, kv8, st);
  else launch_decode_dc<T, D, U, false>(q, kb, vb, o, p, bs, kv8, st);
  return true;
}

template <typename T>
static int launch_decode(const void* q_, const void* kb_, const void* vb_, void* o_,
                         const DecodeParams& p_in, int64_t bs, int D, int Dv, bool kv8, hipStream_t st) {
  DecodeParams p = p_in;
  const T* q = (const T*)q_;
  const T* kb = (const T*)kb_;
  const T* vb = (const T*)vb_;
  T* o = (T*)o_;
  const bool vec = D == Dv && aligned16(q) && aligned16(kb) && aligned16(vb) &&
                   p.q_stride % V16<T>::n == 0;
  constexpr int VE = V16<T>::n;
  // only the matrix-core kernel (head_dim 128, 16-bit) merges the splits itself
  if (!(vec && D == 128 && !std::is_same<T, float>::value && (kv8 || g_decode_variant == 0) &&
        p.num_splits > 1 && p.num_splits <= MERGE_MAX_SPLITS))
    p.tickets = nullptr;
  // an fp8 cache outside the matrix-core kernel's shape (head_dim 128, 16-bit, aligned): the wave-per-head kernel
  const bool kv8_generic = kv8 && !(vec && D == 128 && !std::is_same<T, float>::value);
  if (kv8_generic && (D > 1024 || Dv > 1024)) {
    set_error("decode_attention: head_dim %d / v_head_dim %d > 1024 unsupported", D, Dv);
    return MI355_ERR_UNSUPPORTED;
  }
  if (kv8_generic) {
    hipLaunchKernelGGL((decode_attn_generic_kernel<T>), dim3((unsigned)p.hq, (unsigned)p.num_splits, (unsigned)bs),
                       dim3(64), 0, st, q, kb, vb, o, p, D, Dv, 1);
  } else if (vec && D == 64) launch_decode_d<T, 64, 4>(q, kb, vb, o, p, bs, kv8, st);
  else if (vec && D == 128) launch_decode_d<T, 128, 4>(q, kb, vb, o, p, bs, kv8, st);
  else if (vec && D == 256 && D / VE <= 64) launch_decode_d<T, 256, 2>(q, kb, vb, o, p, bs, kv8, st);
  else {
    if (D > 1024 || Dv > 1024) {
      set_error("decode_attention: head_dim %d / v_head_dim %d > 1024", D, Dv);
      return MI355_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((decode_attn_generic_kernel<T>), dim3((unsigned)p.hq, (unsigned)p.num_splits, (unsigned)bs),
                       dim3(64), 0, st, q, kb, vb, o, p, D, Dv);
  }
  if (p.num_splits > 1 && !p.tickets)
    hipLaunchKernelGGL((decode_merge_kernel<T>), dim3((unsigned)(bs * p.hq)), dim3(128), 0, st,
                       p.part_o, p.part_lse, o, p.num_splits, Dv);
  return MI355_OK;
}

template <typename T>
static int launch_decode_fused(const T* kb, const T* vb, T* o, const DecodeParams& p_in, const FusedQkv& fz,
                               int64_t bs, bool kv8, hipStream_t st) {
  DecodeParams p = p_in;
  if (p.num_splits <= 1 || p.num_splits > MERGE_MAX_SPLITS) p.tickets = nullptr;
  const int G = p.hq / p.hkv;
  const dim3 grid((unsigned)p.hkv, (unsigned)p.num_splits, (unsigned)bs), block(256);
  if (kv8) {
    switch (G) {
      case 8: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 8, false, true, true, true>), grid, block, 0, st, (const T*)nullptr, kb, vb, o, p, fz); break;
      case 4: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 4, false, true, true, true>), grid, block, 0, st, (const T*)nullptr, kb, vb, o, p, fz); break;
      case 2: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 2, false, true, true, true>), grid, block, 0, st, (const T*)nullptr, kb, vb, o, p, fz); break;
      case 1: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 1, false, true, true, true>), grid, block, 0, st, (const T*)nullptr, kb, vb, o, p, fz); break;
      default:
        set_error("decode_attention_fused_qkv: %d query heads per kv head (needs 1, 2, 4 or 8)", G);
        return MI355_ERR_UNSUPPORTED;
    }
    if (p.num_splits > 1 && !p.tickets)
      hipLaunchKernelGGL((decode_merge_kernel<T>), dim3((unsigned)(bs * p.hq)), dim3(128), 0, st,
                         p.part_o, p.part_lse, o, p.num_splits, 128);
    return MI355_OK;
  }
  switch (G) {
    case 8: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 8, false, true, false, true>), grid, block, 0, st, (const T*)nullptr, kb, vb, o, p, fz); break;
    case 4: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 4, false, true, false, true>), grid, block, 0, st, (const T*)nullptr, kb, vb, o, p, fz); break;
    case 2: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 2, false, true, false, true>), grid, block, 0, st, (const T*)nullptr, kb, vb, o, p, fz); break;
    case 1: hipLaunchKernelGGL((decode_attn_mfma_pv_kernel<T, 1, false, true, false, true>), grid, block, 0, st, (const T*)nullptr, kb, vb, o, p, fz); break;
    default:
      set_error("decode_attention_fused_qkv: %d query heads per kv head (needs 1, 2, 4 or 8)", G);
      return MI355_ERR_UNSUPPORTED;
  }
  if (p.num_splits > 1 && !p.tickets)
    hipLaunchKernelGGL((decode_merge_kernel<T>), dim3((unsigned)(bs * p.hq)), dim3(128), 0, st,
                       p.part_o, p.part_lse, o, p.num_splits, 128);
  return MI355_OK;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_decode_num_splits(int64_t bs, int hkv, int64_t max_seq_len, int max_splits) {
  if (bs <= 0 || hkv <= 0 || max_splits <= 1) return 1;
  // Split only until the launch can saturate HBM, and no further: a workgroup of the MFMA kernel
  // streams ~45 GB/s, so ~112 of them already draw what the chip delivers to this access pattern,
  // and every extra split costs fp32 partial traffic plus ~1 us of the merge launch.  Measured at
  // 1064 keys, head_dim 128 (microseconds, best in brackets): 8 workgroups unsplit 17.6 [6-8 splits:
  // 11.6]; 64: 19.9 [3: 14.4; 8: 19.1]; 128: 20.5 [2: 19.7; 4: 22.3]; 256: 31.7 [1; 2: 34.4].
  // Long contexts (>= 1024 keys per split left) with fewer than 192 workgroups are split until every CU has one.
  // 64 sequences x one kv head x 4136 keys with the merge inside the launch: 34.9 / 27.7 / 29.5 / 36.1 / 33.4 /
  // 46.1 us at 2 / 3 / 4 / 6 / 8 / 16 splits -- 192 workgroups of 1 379 keys beat 256 of 1 034 (every split pays its
  // own address chain and one more partial), and the same bytes unsplit (256 sequences x 1 034 keys) take 25.6.
  const int64_t wgs = bs * hkv;
  int64_t want = wgs >= 112 ? 1 : (160 + wgs - 1) / wgs;
  if (want > 1 && wgs * want < 192 && max_seq_len / ((256 + wgs - 1) / wgs) >= 1024) want = (256 + wgs - 1) / wgs;
  // A handful of workgroups (one sequence on a tensor-parallel rank): the launch is a latency chain, ~9.5 ns per key
  // of a split plus ~0.11 us per split merged, least at sqrt(keys / 12) splits.  One sequence x one kv head x 4104
  // keys: 22.3 / 17.6 / 16.2 / 16.2 / 16.6 / 17.5 / 21.7 us at 4 / 8 / 12 / 16 / 24 / 32 / 64 splits.
  int64_t by_len = max_seq_len / 128 > 0 ? max_seq_len / 128 : 1;
  const int64_t by_merge = (int64_t)std::ceil(std::sqrt((double)max_seq_len / 12.0));
  if (by_len > by_merge) by_len = by_merge > 1 ? by_merge : 1;
  if (want > by_len) want = by_len;
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  return (int)want;
}

int mi355_decode_attention(const void* q, const void* k_buffer, const void* v_buffer, void* o,
                           const int32_t* page_table, int mode, const int64_t* req_pool_indices,
                           const int64_t* seq_lens, const int32_t* kv_indptr,
                           int64_t table_stride, float* attn_logits, float* attn_lse,
                           uint32_t* split_tickets, const int32_t* kv_chunk_keys, int num_splits, int64_t bs,
                           int hq, int hkv, int head_dim, int v_head_dim, int64_t q_stride, int64_t num_slots,
                           float sm_scale, float logit_cap, int dtype, void* stream) {
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
  MI355_REQUIRE(num_splits >= 1 && num_splits <= MI355_MAX_KV_SPLITS, MI355_ERR_INVALID,
                "decode_attention: num_splits %d out of range", num_splits);
  MI355_REQUIRE(num_splits == 1 || (attn_logits && attn_lse), MI355_ERR_INVALID,
                "decode_attention: split-KV needs attn_logits and attn_lse scratch");
  MI355_REQUIRE(bs <= 65535, MI355_ERR_UNSUPPORTED, "decode_attention: batch > 65535");
  MI355_REQUIRE(!(dtype & MI355_KV_INTERLEAVED) || ((dtype & MI355_KV_HEAD_MAJOR) && head_dim == v_head_dim), MI355_ERR_INVALID,
                "decode_attention: an interleaved cache is head-major with head_dim == v_head_dim");
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
  p.tickets = split_tickets;
  p.chunk_keys = kv_chunk_keys;
  p.mode = mode;
  p.num_splits = num_splits;
  p.hq = hq;
  p.hkv = hkv;
  p.sm_scale = sm_scale;
  p.logit_cap = logit_cap;
  kv_strides(dtype, num_slots, hkv, head_dim, p.k_slot_stride, p.k_head_stride);
  kv_strides(dtype, num_slots, hkv, v_head_dim, p.v_slot_stride, p.v_head_stride);
  int rc = MI355_OK;
  const bool kv8 = (dtype & MI355_KV_FP8) != 0;
  dtype &= 0xff;
  MI355_DISPATCH_DTYPE(dtype, "decode_attention",
                       rc = launch_decode<T>(q, k_buffer, v_buffer, o, p, bs, head_dim, v_head_dim, kv8,
                                             (hipStream_t)stream));
  if (rc != MI355_OK) return rc;
  MI355_CHECK_LAUNCH("decode_attention");
  return MI355_OK;
}

// ---- split-KV decode attention for callers that cannot allocate -----------------------------
// The partial outputs and log-sum-exps of the KV splits live in a file-scope device array instead
// of caller memory.  `g_static_scratch` is only a host-side stand-in for the symbol: the device
// address comes from hipGetSymbolAddress, once per device (every GPU holds its own copy of the
// array at its own address), and that address is what the kernels receive.
static constexpr int64_t kStaticScratchFloats = MI355_DECODE_STATIC_SCRATCH_BYTES / 4;
static __device__ __align__(256) float g_static_scratch[kStaticScratchFloats];

static constexpr int kMaxDevices = 64;
static std::mutex g_static_mu;
static float* g_static_ptr[kMaxDevices];   // per device; null until that device's first call

// The array lives once per GPU: the copy a launch must use is the one on the device that owns
// `stream` (the calling thread's current device only for the null stream).
static int static_scratch_ptr(hipStream_t stream, float** out) {
  int dev = -1;
  hipError_t e = hipSuccess;
  if (stream != nullptr) {
    hipDevice_t sd = -1;
    e = hipStreamGetDevice(stream, &sd);
    dev = (int)sd;
  } else {
    e = hipGetDevice(&dev);
  }
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    set_error("decode_attention_static: no device for the stream, or device %d outside 0..%d: %s", dev, kMaxDevices - 1,
              hipGetErrorString(e));
    return MI355_ERR_LAUNCH;
  }
  std::lock_guard<std::mutex> lk(g_static_mu);
  if (!g_static_ptr[dev]) {
    // hipGetSymbolAddress answers for the CURRENT device: make the stream's device current for the query
    int cur = -1;
    e = hipGetDevice(&cur);
    if (e == hipSuccess && cur != dev) e = hipSetDevice(dev);
    void* ptr = nullptr;
    if (e == hipSuccess) e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_static_scratch));
    if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
    if (e != hipSuccess || !ptr) {
      set_error("decode_attention_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return MI355_ERR_LAUNCH;
    }
    g_static_ptr[dev] = (float*)ptr;
  }
  *out = g_static_ptr[dev];
  return MI355_OK;
}

int mi355_decode_attention_static(const void* q, const void* k_buffer, const void* v_buffer, void* o,
                                  const int32_t* page_table, int mode, const int64_t* req_pool_indices,
                                  const int64_t* seq_lens, const int32_t* kv_indptr, int64_t table_stride,
                                  int num_splits, int64_t bs, int hq, int hkv, int head_dim, int v_head_dim,
                                  int64_t q_stride, int64_t num_slots, float sm_scale, float logit_cap, int dtype,
                                  void* stream) {
  // bounded before any size arithmetic: the products below then stay far inside int64
  MI355_REQUIRE(bs >= 0 && bs <= 65535 && hq > 0 && hq <= 65535 && v_head_dim > 0 && v_head_dim <= 1024 &&
                    num_splits >= 1 && num_splits <= MI355_MAX_KV_SPLITS,
                MI355_ERR_INVALID, "decode_attention_static: bad shape bs=%lld hq=%d Dv=%d splits=%d", (long long)bs, hq,
                v_head_dim, num_splits);
  // partial outputs [bs, hq, splits, Dv] followed by log-sum-exps [bs, hq, splits], the latter
  // starting on a 256-byte boundary
  const int64_t n_o = bs * hq * (int64_t)num_splits * v_head_dim;
  const int64_t n_o_pad = (n_o + 63) / 64 * 64;
  const int64_t n_lse = bs * hq * (int64_t)num_splits;
  MI355_REQUIRE(num_splits == 1 || n_o_pad + n_lse <= kStaticScratchFloats, MI355_ERR_UNSUPPORTED,
                "decode_attention_static: bs %lld x hq %d x splits %d x (Dv %d + 1) floats = %lld bytes of split "
                "scratch exceed the built-in %lld bytes; use fewer splits or mi355_decode_attention",
                (long long)bs, hq, num_splits, v_head_dim, (long long)((n_o_pad + n_lse) * 4),
                (long long)MI355_DECODE_STATIC_SCRATCH_BYTES);
  float* base = nullptr;
  if (num_splits > 1) {
    const int rc = static_scratch_ptr((hipStream_t)stream, &base);
    if (rc != MI355_OK) return rc;
  }
  // Every (sequence, head, split) cell is written by stage 1 before stage 2 reads it (an empty
  // split stores lse = -inf), so nothing in the array has to start at zero.
  return mi355_decode_attention(q, k_buffer, v_buffer, o, page_table, mode, req_pool_indices, seq_lens, kv_indptr,
                                table_stride, base, base ? base + n_o_pad : nullptr, nullptr, nullptr, num_splits, bs, hq, hkv,
                                head_dim, v_head_dim, q_stride, num_slots, sm_scale, logit_cap, dtype, stream);
}

int mi355_decode_attention_fused_qkv(const float* qkv_part, int nsplit, const float* part_x_scale,
                                     const float* part_w_scale, const int64_t* positions,
                                     const int64_t* out_cache_loc, const void* cos_sin_cache, int cache_dtype, void* k_buffer,
                                     void* v_buffer, void* o, const int32_t* page_table, int mode,
                                     const int64_t* req_pool_indices, const int64_t* seq_lens,
                                     const int32_t* kv_indptr, int64_t table_stride, float* attn_logits,
                                     float* attn_lse, uint32_t* split_tickets, const int32_t* kv_chunk_keys,
                                     int num_splits, int64_t bs, int hq, int hkv, int head_dim, int64_t num_slots,
                                     float sm_scale, int dtype, void* stream) {
  MI355_REQUIRE(qkv_part && positions && out_cache_loc && cos_sin_cache && k_buffer && v_buffer && o && page_table,
                MI355_ERR_INVALID, "decode_attention_fused_qkv: null pointer");
  MI355_REQUIRE(mode == 0 || mode == 1, MI355_ERR_INVALID, "decode_attention_fused_qkv: mode must be 0|1");
  if (mode == 0)
    MI355_REQUIRE(req_pool_indices && seq_lens && table_stride > 0, MI355_ERR_INVALID,
                  "decode_attention_fused_qkv: mode 0 needs req_pool_indices, seq_lens, table_stride");
  else
    MI355_REQUIRE(kv_indptr, MI355_ERR_INVALID, "decode_attention_fused_qkv: mode 1 needs kv_indptr");
  MI355_REQUIRE(bs >= 0 && bs <= 65535 && hq > 0 && hkv > 0 && hq % hkv == 0 && num_slots > 0 && nsplit >= 1 &&
                    nsplit <= 64,
                MI355_ERR_INVALID, "decode_attention_fused_qkv: bad shape hq=%d hkv=%d nsplit=%d", hq, hkv, nsplit);
  const bool kv8 = (dtype & MI355_KV_FP8) != 0;
  const int kv_flags = dtype;
  dtype &= 0xff;
  MI355_REQUIRE(head_dim == 128 && (dtype == MI355_BF16 || dtype == MI355_F16), MI355_ERR_UNSUPPORTED,
                "decode_attention_fused_qkv: needs head_dim 128 and a 16-bit dtype");
  MI355_REQUIRE(cache_dtype == MI355_F32 || cache_dtype == dtype, MI355_ERR_UNSUPPORTED,
                "decode_attention_fused_qkv: cos_sin_cache must be fp32 or the tensor dtype");
  MI355_REQUIRE(num_splits >= 1 && num_splits <= MI355_MAX_KV_SPLITS && (num_splits == 1 || (attn_logits && attn_lse)),
                MI355_ERR_INVALID, "decode_attention_fused_qkv: bad num_splits / missing scratch");
  MI355_REQUIRE(aligned16(qkv_part) && aligned16(k_buffer) && aligned16(v_buffer) && aligned16(part_w_scale),
                MI355_ERR_INVALID, "decode_attention_fused_qkv: misaligned pointer");
  MI355_REQUIRE((part_x_scale == nullptr) == (part_w_scale == nullptr), MI355_ERR_INVALID,
                "decode_attention_fused_qkv: the fp8 scales come in pairs");
  if (bs == 0) return MI355_OK;
  DecodeParams p;
  p.page_table = page_table;
  p.req_pool_indices = req_pool_indices;
  p.seq_lens = seq_lens;
  p.kv_indptr = kv_indptr;
  p.table_stride = table_stride;
  p.q_stride = (int64_t)hq * head_dim;
  p.part_o = attn_logits;
  p.part_lse = attn_lse;
  p.tickets = split_tickets;
  p.chunk_keys = kv_chunk_keys;
  p.mode = mode;
  p.num_splits = num_splits;
  p.hq = hq;
  p.hkv = hkv;
  p.sm_scale = sm_scale;
  p.logit_cap = 0.f;
  kv_strides(kv_flags, num_slots, hkv, head_dim, p.k_slot_stride, p.k_head_stride);
  kv_strides(kv_flags, num_slots, hkv, head_dim, p.v_slot_stride, p.v_head_stride);
  FusedQkv fz;
  fz.part = qkv_part;
  fz.nsplit = nsplit;
  fz.xs = part_x_scale;
  fz.ws = part_w_scale;
  fz.row = (int64_t)(hq + 2 * hkv) * head_dim;
  fz.split_stride = bs * fz.row;
  fz.positions = positions;
  fz.out_loc = out_cache_loc;
  fz.cos_sin = cos_sin_cache;
  fz.cache_f32 = cache_dtype == MI355_F32;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dtype == MI355_BF16) rc = launch_decode_fused<bf16_t>((const bf16_t*)k_buffer, (const bf16_t*)v_buffer, (bf16_t*)o, p, fz, bs, kv8, st);
  else rc = launch_decode_fused<f16_t>((const f16_t*)k_buffer, (const f16_t*)v_buffer, (f16_t*)o, p, fz, bs, kv8, st);
  if (rc != MI355_OK) return rc;
  MI355_CHECK_LAUNCH("decode_attention_fused_qkv");
  return MI355_OK;
}

static int decode_attention_mla_impl(const MlaRope* rope, const void* q, const void* kv_buffer, void* o,
                               const int32_t* page_table, int mode,
                               const int64_t* req_pool_indices, const int64_t* seq_lens,
                               const int32_t* kv_indptr, int64_t table_stride, float* attn_logits,
                               float* attn_lse, uint32_t* split_tickets, const int32_t* kv_chunk_keys,
                               int num_splits, int64_t bs, int hq, int kv_lora_rank, int rope_dim, int64_t q_stride,
                               int64_t num_slots, float sm_scale, float logit_cap, int dtype, void* stream) {
  MI355_REQUIRE(q && kv_buffer && o && page_table, MI355_ERR_INVALID, "decode_attention_mla: null pointer");
  MI355_REQUIRE(mode == 0 || mode == 1, MI355_ERR_INVALID, "decode_attention_mla: mode must be 0|1");
  if (mode == 0)
    MI355_REQUIRE(req_pool_indices && seq_lens && table_stride > 0, MI355_ERR_INVALID,
                  "decode_attention_mla: mode 0 needs req_pool_indices, seq_lens, table_stride");
  else
    MI355_REQUIRE(kv_indptr, MI355_ERR_INVALID, "decode_attention_mla: mode 1 needs kv_indptr");
  const int D = kv_lora_rank + rope_dim;
  const bool kv8 = (dtype & MI355_KV_FP8) != 0;   // the latent cache holds e4m3fn bytes
  dtype &= 0xff;
  MI355_REQUIRE(bs >= 0 && bs <= 65535 && hq > 0 && kv_lora_rank > 0 && rope_dim >= 0 && num_slots > 0 &&
                    q_stride >= (int64_t)hq * D,
                MI355_ERR_INVALID, "decode_attention_mla: bad shape hq=%d rank=%d rope=%d", hq, kv_lora_rank, rope_dim);
  MI355_REQUIRE(D <= 1024, MI355_ERR_UNSUPPORTED, "decode_attention_mla: latent row of %d > 1024", D);
  MI355_REQUIRE(num_splits >= 1 && num_splits <= MI355_MAX_KV_SPLITS && (num_splits == 1 || (attn_logits && attn_lse)),
                MI355_ERR_INVALID, "decode_attention_mla: bad num_splits / missing scratch");
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
  p.hkv = 1;
  p.sm_scale = sm_scale;
  p.logit_cap = logit_cap;
  p.chunk_keys = kv_chunk_keys;
  p.k_slot_stride = D;   // one latent row per slot; the value is its first kv_lora_rank features
  p.v_slot_stride = D;
  p.k_head_stride = p.v_head_stride = D;
  hipStream_t st = (hipStream_t)stream;
  if (rope) {
    // rotation + cache write inside the launch: only the LDS-ring kernel has it
    MI355_REQUIRE(kv_lora_rank == MLA_RANK && rope_dim == MLA_ROPE && (dtype == MI355_BF16 || dtype == MI355_F16) &&
                      aligned16(q) && aligned16(kv_buffer) && q_stride % 8 == 0 && aligned16(attn_logits) && aligned16(o) &&
                      g_decode_variant == 0,
                  MI355_ERR_UNSUPPORTED,
                  "decode_attention_mla_fused_rope: needs rank 512 + rope 64, a 16-bit dtype, a 16-bit or e4m3fn cache, 16-byte aligned pointers");
  }
  if (kv8) {
    // e4m3fn latent rows: the fp8 LDS ring (raw rows in LDS, widened per fragment); the register kernel (widening
    // on use, merge launch) when a pointer is not 16-byte aligned or as development variant 2
    MI355_REQUIRE(kv_lora_rank == MLA_RANK && rope_dim == MLA_ROPE && (dtype == MI355_BF16 || dtype == MI355_F16) &&
                      aligned16(q) && (reinterpret_cast<uintptr_t>(kv_buffer) & 7) == 0 && q_stride % 8 == 0,
                  MI355_ERR_UNSUPPORTED,
                  "decode_attention_mla: an fp8 latent cache needs rank 512 + rope 64, a 16-bit dtype and aligned pointers");
    const dim3 grid((unsigned)((hq + 15) / 16), (unsigned)num_splits, (unsigned)bs);
    if (g_decode_variant == 0 && aligned16(kv_buffer) && aligned16(attn_logits) && aligned16(o)) {
      constexpr int NSB_ = 3;
      // two KV splits per workgroup (two teams of four waves) when the halved grid still gives every CU a workgroup
      const bool two = num_splits % 2 == 0 && (int64_t)grid.x * grid.y * grid.z >= 512 && g_mla_teams != 1;
      const int s_out = two ? num_splits / 2 : num_splits;
      const dim3 grid_t(grid.x, (unsigned)s_out, grid.z);
      if (s_out > 1 && s_out <= MERGE_MAX_SPLITS) p.tickets = split_tickets;   // merged inside the launch
#define MLR8_GO(T_, KERN_, TEAMS_)                                                                                \
  do {                                                                                                         \
    auto kern = KERN_;                                                                                         \
    const int lds_ = TEAMS_ * m8_lds_bytes(NSB_);                                                              \
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_);            \
    hipLaunchKernelGGL(kern, grid_t, dim3(256 * TEAMS_), lds_, st, (const T_*)q, (const uint8_t*)kv_buffer,    \
                       (T_*)o, p, rope ? *rope : MlaRope{});                                                   \
  } while (0)
#define MLR8_LAUNCH(T_, CAP_)                                                                                  \
  do {                                                                                                         \
    if (two) {                                                                                                 \
      if (rope) MLR8_GO(T_, (decode_attn_mla_ring8_kernel<T_, CAP_, NSB_, true, 2>), 2);                           \
      else MLR8_GO(T_, (decode_attn_mla_ring8_kernel<T_, CAP_, NSB_, false, 2>), 2);                               \
    } else {                                                                                                   \
      if (rope) MLR8_GO(T_, (decode_attn_mla_ring8_kernel<T_, CAP_, NSB_, true, 1>), 1);                           \
      else MLR8_GO(T_, (decode_attn_mla_ring8_kernel<T_, CAP_, NSB_, false, 1>), 1);                               \
    }                                                                                                          \
    if (s_out > 1 && !p.tickets)                                                                               \
      hipLaunchKernelGGL((decode_merge_kernel<T_>), dim3((unsigned)(bs * hq)), dim3(128), 0, st, p.part_o,     \
                         p.part_lse, (T_*)o, s_out, kv_lora_rank);                                             \
  } while (0)
      if (dtype == MI355_BF16) { if (logit_cap > 0.f) MLR8_LAUNCH(bf16_t, true); else MLR8_LAUNCH(bf16_t, false); }
      else { if (logit_cap > 0.f) MLR8_LAUNCH(f16_t, true); else MLR8_LAUNCH(f16_t, false); }
#undef MLR8_GO
#undef MLR8_LAUNCH
      MI355_CHECK_LAUNCH("decode_attention_mla");
      return MI355_OK;
    }
#define MLA8_LAUNCH(T_, CAP_)                                                                                  \
  do {                                                                                                         \
    auto kern = decode_attn_mla_mfma_kernel<T_, CAP_, true>;                                                   \
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, MLA_LDS_BYTES);   \
    hipLaunchKernelGGL(kern, grid, dim3(256), MLA_LDS_BYTES, st, (const T_*)q, (const T_*)kv_buffer, (T_*)o, p); \
    if (num_splits > 1)                                                                                        \
      hipLaunchKernelGGL((decode_merge_kernel<T_>), dim3((unsigned)(bs * hq)), dim3(128), 0, st, p.part_o,     \
                         p.part_lse, (T_*)o, num_splits, kv_lora_rank);                                        \
  } while (0)
    if (dtype == MI355_BF16) { if (logit_cap > 0.f) MLA8_LAUNCH(bf16_t, true); else MLA8_LAUNCH(bf16_t, false); }
    else { if (logit_cap > 0.f) MLA8_LAUNCH(f16_t, true); else MLA8_LAUNCH(f16_t, false); }
#undef MLA8_LAUNCH
    MI355_CHECK_LAUNCH("decode_attention_mla");
    return MI355_OK;
  }
  if (kv_lora_rank == MLA_RANK && rope_dim == MLA_ROPE && (dtype == MI355_BF16 || dtype == MI355_F16) &&
      aligned16(q) && aligned16(kv_buffer) && q_stride % 8 == 0 && (g_decode_variant == 0 || g_decode_variant == 2 || g_decode_variant == 5)) {
    const dim3 grid((unsigned)((hq + 15) / 16), (unsigned)num_splits, (unsigned)bs);
    if ((g_decode_variant == 0 || g_decode_variant == 5) && aligned16(attn_logits)) {   // rows through the LDS ring (2 = the register kernel, 5 = the deep ring, for ablation)
      constexpr int NSB_ = 2;
      const bool deep = g_decode_variant == 5;
      // two KV splits per workgroup (two teams of four waves) when the halved grid still gives every CU a workgroup
      const bool two = !deep && num_splits % 2 == 0 && (int64_t)grid.x * grid.y * grid.z >= 512 && g_mla_teams != 1;
      const int s_out = two ? num_splits / 2 : num_splits;
      const dim3 grid_t(grid.x, (unsigned)s_out, grid.z);
      if (s_out > 1 && s_out <= MERGE_MAX_SPLITS && aligned16(o)) p.tickets = split_tickets;   // this kernel merges the splits itself
#define MLR_GO(T_, KERN_, LDS_, THREADS_)                                                                      \
  do {                                                                                                         \
    auto kern = KERN_;                                                                                         \
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_);            \
    hipLaunchKernelGGL(kern, grid_t, dim3(THREADS_), LDS_, st, (const T_*)q, (const T_*)kv_buffer, (T_*)o, p,  \
                       rope ? *rope : MlaRope{});                                                              \
  } while (0)
#define MLR_LAUNCH(T_, CAP_)                                                                                   \
  do {                                                                                                         \
    if (two && rope) MLR_GO(T_, (decode_attn_mla_ring_kernel<T_, CAP_, NSB_, true, 2>), 2 * mlr_lds_bytes(NSB_), 512); \
    else if (two) MLR_GO(T_, (decode_attn_mla_ring_kernel<T_, CAP_, NSB_, false, 2>), 2 * mlr_lds_bytes(NSB_), 512);   \
    else if (rope) MLR_GO(T_, (decode_attn_mla_ring_kernel<T_, CAP_, NSB_, true>), mlr_lds_bytes(NSB_), 256);  \
    else if (deep) MLR_GO(T_, (decode_attn_mla_ring_kernel<T_, CAP_, 4>), mlr_lds_bytes(4), 256);              \
    else MLR_GO(T_, (decode_attn_mla_ring_kernel<T_, CAP_, NSB_>), mlr_lds_bytes(NSB_), 256);                  \
    if (s_out > 1 && !p.tickets)                                                                               \
      hipLaunchKernelGGL((decode_merge_kernel<T_>), dim3((unsigned)(bs * hq)), dim3(128), 0, st, p.part_o,     \
                         p.part_lse, (T_*)o, s_out, kv_lora_rank);                                             \
  } while (0)
      if (dtype == MI355_BF16) { if (logit_cap > 0.f) MLR_LAUNCH(bf16_t, true); else MLR_LAUNCH(bf16_t, false); }
      else { if (logit_cap > 0.f) MLR_LAUNCH(f16_t, true); else MLR_LAUNCH(f16_t, false); }
#undef MLR_GO
#undef MLR_LAUNCH
      MI355_CHECK_LAUNCH("decode_attention_mla");
      return MI355_OK;
    }
#define MLA_LAUNCH(T_, CAP_)                                                                                   \
  do {                                                                                                         \
    auto kern = decode_attn_mla_mfma_kernel<T_, CAP_>;                                                         \
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, MLA_LDS_BYTES);   \
    hipLaunchKernelGGL(kern, grid, dim3(256), MLA_LDS_BYTES, st, (const T_*)q, (const T_*)kv_buffer, (T_*)o, p); \
    if (num_splits > 1)                                                                                        \
      hipLaunchKernelGGL((decode_merge_kernel<T_>), dim3((unsigned)(bs * hq)), dim3(128), 0, st, p.part_o,     \
                         p.part_lse, (T_*)o, num_splits, kv_lora_rank);                                        \
  } while (0)
    if (dtype == MI355_BF16) { if (logit_cap > 0.f) MLA_LAUNCH(bf16_t, true); else MLA_LAUNCH(bf16_t, false); }
    else { if (logit_cap > 0.f) MLA_LAUNCH(f16_t, true); else MLA_LAUNCH(f16_t, false); }
#undef MLA_LAUNCH
    MI355_CHECK_LAUNCH("decode_attention_mla");
    return MI355_OK;
  }
  MI355_DISPATCH_DTYPE(dtype, "decode_attention_mla", {
    hipLaunchKernelGGL((decode_attn_generic_kernel<T>), dim3((unsigned)hq, (unsigned)num_splits, (unsigned)bs),
                       dim3(64), 0, st, (const T*)q, (const T*)kv_buffer, (const T*)kv_buffer, (T*)o, p, D,
                       kv_lora_rank);
    if (num_splits > 1)
      hipLaunchKernelGGL((decode_merge_kernel<T>), dim3((unsigned)(bs * hq)), dim3(128), 0, st, p.part_o,
                         p.part_lse, (T*)o, num_splits, kv_lora_rank);
  });
  MI355_CHECK_LAUNCH("decode_attention_mla");
  return MI355_OK;
}

int mi355_decode_attention_mla(const void* q, const void* kv_buffer, void* o,
                               const int32_t* page_table, int mode,
                               const int64_t* req_pool_indices, const int64_t* seq_lens,
                               const int32_t* kv_indptr, int64_t table_stride, float* attn_logits,
                               float* attn_lse, uint32_t* split_tickets, const int32_t* kv_chunk_keys,
                               int num_splits, int64_t bs, int hq, int kv_lora_rank, int rope_dim, int64_t q_stride,
                               int64_t num_slots, float sm_scale, float logit_cap, int dtype, void* stream) {
  return decode_attention_mla_impl(nullptr, q, kv_buffer, o, page_table, mode, req_pool_indices, seq_lens, kv_indptr,
                                   table_stride, attn_logits, attn_lse, split_tickets, kv_chunk_keys, num_splits, bs, hq,
                                   kv_lora_rank, rope_dim, q_stride, num_slots, sm_scale, logit_cap, dtype, stream);
}

int mi355_decode_attention_mla_fused_rope_supported(int kv_lora_rank, int rope_dim, int dtype, int cache_is_fp8) {
  (void)cache_is_fp8;   // 16-bit and e4m3fn caches both have an LDS-ring kernel with the rotation inside
  return kv_lora_rank == MLA_RANK && rope_dim == MLA_ROPE && (dtype == MI355_BF16 || dtype == MI355_F16) &&
         g_decode_variant == 0;
}

int mi355_decode_attention_mla_fused_rope(const void* q, const void* k_nope, int64_t k_nope_stride, const void* k_pe,
                                          int64_t k_pe_stride, const int64_t* positions, const int64_t* out_cache_loc,
                                          const void* cos_sin_cache, int cache_dtype, int is_neox, void* kv_buffer,
                                          void* o, const int32_t* page_table, int mode,
                                          const int64_t* req_pool_indices, const int64_t* seq_lens,
                                          const int32_t* kv_indptr, int64_t table_stride, float* attn_logits,
                                          float* attn_lse, uint32_t* split_tickets, const int32_t* kv_chunk_keys,
                                          int num_splits, int64_t bs, int hq, int kv_lora_rank, int rope_dim,
                                          int64_t q_stride, int64_t num_slots, float sm_scale, float logit_cap,
                                          int dtype, void* stream) {
  MI355_REQUIRE(k_nope && k_pe && positions && out_cache_loc && cos_sin_cache, MI355_ERR_INVALID,
                "decode_attention_mla_fused_rope: null pointer");
  MI355_REQUIRE(k_nope_stride >= kv_lora_rank && k_nope_stride % 8 == 0 && k_pe_stride >= rope_dim && k_pe_stride % 8 == 0 &&
                    aligned16(k_nope) && aligned16(k_pe),
                MI355_ERR_INVALID, "decode_attention_mla_fused_rope: k_nope / k_pe rows must be 16-byte aligned");
  MI355_REQUIRE(cache_dtype == MI355_F32 || cache_dtype == (dtype & 0xff), MI355_ERR_UNSUPPORTED,
                "decode_attention_mla_fused_rope: cos_sin_cache must be fp32 or the tensor dtype");
  MlaRope rp;
  rp.k_nope = k_nope;
  rp.k_pe = k_pe;
  rp.k_nope_stride = k_nope_stride;
  rp.k_pe_stride = k_pe_stride;
  rp.positions = positions;
  rp.out_loc = out_cache_loc;
  rp.cos_sin = cos_sin_cache;
  rp.cache_f32 = cache_dtype == MI355_F32;
  rp.is_neox = is_neox != 0;
  return decode_attention_mla_impl(&rp, q, kv_buffer, o, page_table, mode, req_pool_indices, seq_lens, kv_indptr,
                                   table_stride, attn_logits, attn_lse, split_tickets, kv_chunk_keys, num_splits, bs, hq,
                                   kv_lora_rank, rope_dim, q_stride, num_slots, sm_scale, logit_cap, dtype, stream);
}

int mi355_extend_attention_mla_supported(int hq, int kv_lora_rank, int rope_dim, int dtype) {
  return hq > 0 && kv_lora_rank == MLA_RANK && rope_dim == MLA_ROPE && (dtype == MI355_BF16 || dtype == MI355_F16);
}

int mi355_extend_attention_mla(const void* q, const void* kv_buffer, void* o, const int32_t* req_to_token,
                               const int64_t* req_pool_indices, const int64_t* seq_lens,
                               const int32_t* qo_indptr, int64_t table_stride, int64_t bs,
                               int64_t max_extend_len, int hq, int kv_lora_rank, int rope_dim,
                               int64_t q_stride, int64_t num_slots, float sm_scale, float logit_cap,
                               int dtype, void* stream) {
  MI355_REQUIRE(q && kv_buffer && o && req_to_token && req_pool_indices && seq_lens && qo_indptr, MI355_ERR_INVALID,
                "extend_attention_mla: null pointer");
  MI355_REQUIRE(mi355_extend_attention_mla_supported(hq, kv_lora_rank, rope_dim, dtype), MI355_ERR_UNSUPPORTED,
                "extend_attention_mla: needs a latent row of 512 + 64 features and a 16-bit dtype (rank=%d rope=%d)",
                kv_lora_rank, rope_dim);
  MI355_REQUIRE(bs >= 0 && bs <= 65535 && max_extend_len >= 0 && max_extend_len <= 4 * 65535 && table_stride > 0 &&
                    num_slots > 0 && q_stride >= (int64_t)hq * MLA_D && q_stride % 8 == 0,
                MI355_ERR_INVALID, "extend_attention_mla: bad shape bs=%ld max_extend_len=%ld hq=%d", (long)bs,
                (long)max_extend_len, hq);
  MI355_REQUIRE(aligned16(q) && aligned16(kv_buffer) && aligned16(o), MI355_ERR_INVALID,
                "extend_attention_mla: pointers must be 16-byte aligned");
  if (bs == 0 || max_extend_len == 0) return MI355_OK;
  DecodeParams p{};
  p.page_table = req_to_token;
  p.req_pool_indices = req_pool_indices;
  p.seq_lens = seq_lens;
  p.table_stride = table_stride;
  p.q_stride = q_stride;
  p.mode = 0;
  p.num_splits = 1;
  p.hq = hq;
  p.hkv = 1;
  p.sm_scale = sm_scale;
  p.logit_cap = logit_cap;
  p.k_slot_stride = MLA_D;
  p.v_slot_stride = MLA_D;
  p.k_head_stride = p.v_head_stride = MLA_D;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((hq + 15) / 16), (unsigned)((max_extend_len + MLX_TOKENS - 1) / MLX_TOKENS), (unsigned)bs);
  constexpr int lds = mlr_lds_bytes(4);
#define MLX_LAUNCH(T_, CAP_)                                                                                  \
  do {                                                                                                        \
    auto kern = extend_attn_mla_tile_kernel<T_, CAP_>;                                                        \
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);            \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const T_*)q, (const T_*)kv_buffer, (T_*)o, p, qo_indptr); \
  } while (0)
  if (dtype == MI355_BF16) { if (logit_cap > 0.f) MLX_LAUNCH(bf16_t, true); else MLX_LAUNCH(bf16_t, false); }
  else { if (logit_cap > 0.f) MLX_LAUNCH(f16_t, true); else MLX_LAUNCH(f16_t, false); }
#undef MLX_LAUNCH
  MI355_CHECK_LAUNCH("extend_attention_mla");
  return MI355_OK;
}

}  // extern "C"
