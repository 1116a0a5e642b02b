// Sampling tail of the step: greedy arg-max, temperature softmax, and sort-free joint
// top-k / top-p / min-p sampling over fp32 [rows, vocab] logits.  One workgroup of 1024
// threads per row; rows are 0.5 MB at vocab 128256, so every pass after the first is served
// from L2 / Infinity Cache.
#include "common.cuh"
#include <math.h>
#include <algorithm>

namespace mi355 {

constexpr int SAMP_THREADS = 1024;

// ---- argmax: first index of the maximum (torch.argmax tie rule) --------------------------
// (maximum, first index) of elements [4 v0, 4 v1) of `row` -- plus the row's tail past the last whole
// float4 when `with_tail` -- reduced over the workgroup; valid in thread 0.  vec = the 16-byte path.
__device__ __forceinline__ void argmax_scan(const float* __restrict__ row, int64_t vocab, bool vec, int64_t v0, int64_t v1,
                                            bool with_tail, float& best_out, int& bidx_out) {
  __shared__ float sv[SAMP_THREADS / 64];
  __shared__ int si[SAMP_THREADS / 64];
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  if (vec) {
    const int64_t nv = v1;
    auto take = [&](const float4& v, int64_t i) {
      const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (a[j] > best || (a[j] == best && (int)(i * 4 + j) < bidx)) {
          best = a[j];
          bidx = (int)(i * 4 + j);
        }
    };
    // A lane meets its elements in rising index order, so "strictly greater" alone keeps the first
    // maximum: three branch-free instructions per element.  It cannot pick among elements equal to
    // the initial -inf; a lane that ends with nothing selected repeats the pass with the full rule.
    auto take_gt = [&](const float4& v, int64_t i) {
      const float a[4] = {v.x, v.y, v.z, v.w};
      const int base = (int)i * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool gt = a[j] > best;
        best = gt ? a[j] : best;
        bidx = gt ? base + j : bidx;
      }
    };
    // sixteen loads in flight per lane before the first comparison: 64 rows are 64 workgroups, so
    // a row is read at the latency of its load rounds (two for a 128k vocabulary), not at bandwidth
    constexpr int U = 16;
    auto scan = [&](auto&& pick) {
      for (int64_t i = v0 + threadIdx.x; i < nv; i += U * SAMP_THREADS) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (i + u * SAMP_THREADS < nv) v[u] = *reinterpret_cast<const float4*>(row + (i + u * SAMP_THREADS) * 4);
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (i + u * SAMP_THREADS < nv) pick(v[u], i + u * SAMP_THREADS);
      }
    };
    scan(take_gt);
    if (bidx == 0x7fffffff) scan(take);
    if (with_tail)
      for (int64_t i = (vocab / 4) * 4 + threadIdx.x; i < vocab; i += SAMP_THREADS)
        if (row[i] > best || (row[i] == best && (int)i < bidx)) {
          best = row[i];
          bidx = (int)i;
        }
  } else {
    for (int64_t i = 4 * v0 + threadIdx.x; i < (with_tail ? vocab : 4 * v1); i += SAMP_THREADS)
      if (row[i] > best || (row[i] == best && (int)i < bidx)) {
        best = row[i];
        bidx = (int)i;
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ov > best || (ov == best && oi < bidx)) {
      best = ov;
      bidx = oi;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = best;
    si[threadIdx.x >> 6] = bidx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < SAMP_THREADS / 64; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bidx)) {
        best = sv[w];
        bidx = si[w];
      }
    best_out = best;
    bidx_out = bidx;
  }
}

__global__ __launch_bounds__(SAMP_THREADS) void argmax_kernel(int64_t* __restrict__ ids,
                                                             const float* __restrict__ logits,
                                                             int64_t vocab, int64_t row_stride) {
  const float* row = logits + (int64_t)blockIdx.x * row_stride;
  const bool vec = (row_stride % 4 == 0) && aligned16(logits);
  float best;
  int bidx;
  argmax_scan(row, vocab, vec, 0, vocab / 4, true, best, bidx);
  // a row without any finite-or-greater element (all NaN / -inf): index 0 like torch on -inf
  if (threadIdx.x == 0) ids[blockIdx.x] = bidx == 0x7fffffff ? 0 : bidx;
}

// A few rows (a decode batch) are a few workgroups: one workgroup reads its 0.5-MB row at what ONE CU can
// draw.  Here a row is cut into gridDim.y chunks; every chunk's (maximum, first index) is folded into the
// row's 64-bit cell with an atomic max -- key = (order-preserving image of the float) << 32 | ~index, so a
// larger key is a larger value or, among equal values, a smaller index: the same winner in any arrival
// order -- and the chunk that draws the row's last ticket decodes the cell into ids[row] and hands cell and
// ticket counter back zeroed.  All traffic on cells and counters is returning agent-scope atomics (each is
// complete before the next is issued), so nothing depends on cache write-back.
// grid (rows, chunks); cells: rows x uint64, tickets: rows x uint32, both zero between launches.
__global__ __launch_bounds__(SAMP_THREADS) void argmax_chunk_kernel(int64_t* __restrict__ ids,
                                                                   const float* __restrict__ logits, int64_t vocab,
                                                                   int64_t row_stride,
                                                                   unsigned long long* __restrict__ cells,
                                                                   unsigned* __restrict__ tickets) {
  const int r = blockIdx.x, c = blockIdx.y, chunks = gridDim.y;
  const float* row = logits + (int64_t)r * row_stride;
  const bool vec = (row_stride % 4 == 0) && aligned16(logits);
  const int64_t nv = vocab / 4, per = (nv + chunks - 1) / chunks;
  const int64_t v0 = c * per < nv ? c * per : nv, v1 = v0 + per < nv ? v0 + per : nv;
  float best;
  int bidx;
  argmax_scan(row, vocab, vec, v0, v1, c == chunks - 1, best, bidx);
  if (threadIdx.x != 0) return;
  unsigned u = __builtin_bit_cast(unsigned, best + 0.f);   // (-0 + 0 = +0: the two zeros are one value, as in a float compare)
  u = (u >> 31) ? ~u : (u | 0x80000000u);
  const unsigned long long key = ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)bidx);
  (void)__hip_atomic_fetch_max(&cells[r], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the maximum is in before the ticket is drawn
  const unsigned t = __hip_atomic_fetch_add(&tickets[r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t != (unsigned)(chunks - 1)) return;
  // every chunk's maximum is in (each was complete before its ticket was drawn): read the cell at the
  // point the atomics work at, then put cell and counter back to zero (nothing waits for those stores)
  const unsigned long long win = __hip_atomic_load(&cells[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&cells[r], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&tickets[r], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned idx = 0xffffffffu - (unsigned)(win & 0xffffffffu);
  ids[r] = idx == 0x7fffffffu ? 0 : (int64_t)idx;
}

// ---- softmax(logits / temperature), in place ----------------------------------------------
__device__ __forceinline__ float block_reduce_max(float x, float* red) {
  x = wave_max(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, red[i]);
  return t;
}

__global__ __launch_bounds__(SAMP_THREADS) void softmax_temperature_kernel(
    float* __restrict__ logits, const float* __restrict__ temperatures, int64_t vocab,
    int64_t row_stride) {
  __shared__ float red[SAMP_THREADS / 64];
  float* row = logits + (int64_t)blockIdx.x * row_stride;
  const float t = temperatures[blockIdx.x];
  float mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < vocab; i += SAMP_THREADS) mx = fmaxf(mx, row[i] / t);
  mx = block_reduce_max(mx, red);
  float sum = 0.f;
  for (int64_t i = threadIdx.x; i < vocab; i += SAMP_THREADS) sum += expf(row[i] / t - mx);
  sum = block_sum(sum, red);
  for (int64_t i = threadIdx.x; i < vocab; i += SAMP_THREADS) row[i] = expf(row[i] / t - mx) / sum;
}

// ---- joint top-k / top-p / min-p filter + inverse-CDF draw --------------------------------
// The kept set is {p >= tau}: tau is the smallest probability value v such that fewer than k
// entries are strictly larger than v and the mass strictly above v is <= top_p.  It is found
// by a 4-level radix select on the float bit pattern (non-negative floats order like their
// bits) with 256-bin count/mass histograms in LDS, i.e. 4 passes over the row instead of a
// sort.  min_p raises tau to min_p * pmax.  The draw walks the kept entries in index order.
__global__ __launch_bounds__(SAMP_THREADS) void topk_topp_minp_sample_kernel(
    int32_t* __restrict__ ids, const float* __restrict__ probs, const int32_t* __restrict__ top_ks,
    const float* __restrict__ top_ps, const float* __restrict__ min_ps,
    const float* __restrict__ uniform, int64_t vocab, int64_t row_stride) {
  __shared__ unsigned int hcount[256];
  __shared__ float hmass[256];
  __shared__ float red[SAMP_THREADS / 64];
  __shared__ unsigned int s_prefix;   // bits fixed so far (high bits of tau)
  __shared__ unsigned int s_count;    // entries strictly above the current bin range
  __shared__ float s_mass;            // mass strictly above the current bin range
  __shared__ float s_scan[SAMP_THREADS];
  __shared__ int s_pick;

  const float* row = probs + (int64_t)blockIdx.x * row_stride;
  const int64_t k = top_ks ? (int64_t)top_ks[blockIdx.x] : (int64_t)1 << 30;
  float top_p = top_ps ? top_ps[blockIdx.x] : 1.0f;
  if (top_p >= 1.0f) top_p = INFINITY;  // disabled: no rounding artefacts at the tail
  const float min_p = min_ps ? min_ps[blockIdx.x] : 0.0f;

  if (threadIdx.x == 0) {
    s_prefix = 0;
    s_count = 0;
    s_mass = 0.f;
  }
  float pmax = 0.f;
  for (int64_t i = threadIdx.x; i < vocab; i += SAMP_THREADS) pmax = fmaxf(pmax, row[i]);
  pmax = block_reduce_max(pmax, red);

  for (int level = 0; level < 4; ++level) {
    const int shift = 24 - 8 * level;
    for (int i = threadIdx.x; i < 256; i += SAMP_THREADS) {
      hcount[i] = 0;
      hmass[i] = 0.f;
    }
    __syncthreads();
    const unsigned int prefix = s_prefix;
    const unsigned int himask = level == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int64_t i = threadIdx.x; i < vocab; i += SAMP_THREADS) {
      const float v = row[i];
      const unsigned int bits = __float_as_uint(v > 0.f ? v : 0.f);
      if ((bits & himask) == prefix) {
        const unsigned int bin = (bits >> shift) & 0xffu;
        atomicAdd(&hcount[bin], 1u);
        atomicAdd(&hmass[bin], v > 0.f ? v : 0.f);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned int cnt = s_count;
      float mass = s_mass;
      int bin = 255;
      // walk down while every entry of the bin is certainly kept
      for (; bin > 0; --bin) {
        if (hcount[bin] == 0) continue;
        // the smallest entry of this bin sees at most cnt + hcount-1 larger entries and at
        // most mass + hmass (minus itself) above it; if the whole bin passes, keep going.
        const bool all_pass = (int64_t)cnt + hcount[bin] <= k && mass + hmass[bin] <= top_p;
        if (!all_pass) break;
        cnt += hcount[bin];
        mass += hmass[bin];
      }
      s_prefix = prefix | ((unsigned int)bin << shift);
      s_count = cnt;
      s_mass = mass;
    }
    __syncthreads();
  }
  // s_prefix is now a full 32-bit pattern v0.  Entries > v0 number s_count with mass s_mass;
  // entries == v0 are kept iff s_count < k and s_mass <= top_p (and v0 > 0).
  const unsigned int v0bits = s_prefix;
  const bool keep_v0 = (int64_t)s_count < k && s_mass <= top_p;
  float tau = __uint_as_float(keep_v0 ? v0bits : v0bits + 1u);  // next float up excludes v0
  tau = fmaxf(tau, min_p * pmax);
  if (!(tau > 0.f)) tau = __uint_as_float(1u);  // never keep zero-probability entries

  // total kept mass
  const int64_t per = (vocab + SAMP_THREADS - 1) / SAMP_THREADS;
  const int64_t lo = threadIdx.x * per, hi = lo + per < vocab ? lo + per : vocab;
  float part = 0.f;
  for (int64_t i = lo; i < hi; ++i) {
    const float v = row[i];
    if (v >= tau) part += v;
  }
  s_scan[threadIdx.x] = part;
  if (threadIdx.x == 0) s_pick = -1;
  __syncthreads();
  if (threadIdx.x == 0) {
    // serial exclusive scan of 1024 partials: tiny, deterministic
    float run = 0.f;
    for (int i = 0; i < SAMP_THREADS; ++i) {
      const float t = s_scan[i];
      s_scan[i] = run;
      run += t;
    }
    red[0] = run;
  }
  __syncthreads();
  const float total = red[0];
  const float target = uniform[blockIdx.x] * total;
  // the chunk containing the target: exclusive prefix <= target < prefix + part
  const float before = s_scan[threadIdx.x];
  if (part > 0.f && before <= target && (target < before + part)) {
    float run = before;
    int pick = -1;
    for (int64_t i = lo; i < hi; ++i) {
      const float v = row[i];
      if (v >= tau) {
        run += v;
        pick = (int)i;
        if (run > target) break;
      }
    }
    atomicMax(&s_pick, pick);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int pick = s_pick;
    if (pick < 0) {
      // rounding at the very end of the CDF: fall back to the last kept entry
      for (int64_t i = vocab - 1; i >= 0; --i)
        if (row[i] >= tau) {
          pick = (int)i;
          break;
        }
      if (pick < 0) pick = 0;
    }
    ids[blockIdx.x] = pick;
  }
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_argmax(int64_t* ids, const float* logits, int64_t rows, int64_t vocab,
                 int64_t row_stride, void* stream) {
  MI355_REQUIRE(ids && logits, MI355_ERR_INVALID, "argmax: null pointer");
  MI355_REQUIRE(rows >= 0 && vocab > 0 && vocab < (1LL << 31) && row_stride >= vocab,
                MI355_ERR_INVALID, "argmax: bad shape");
  if (rows == 0) return MI355_OK;
  hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)rows), dim3(SAMP_THREADS), 0,
                     (hipStream_t)stream, ids, logits, vocab, row_stride);
  MI355_CHECK_LAUNCH("argmax");
  return MI355_OK;
}

int64_t mi355_argmax_workspace_bytes(int64_t rows) { return rows > 0 ? rows * 16 : 0; }

int mi355_argmax_chunked(int64_t* ids, const float* logits, int64_t rows, int64_t vocab, int64_t row_stride,
                         void* workspace, int64_t workspace_bytes, void* stream) {
  MI355_REQUIRE(ids && logits, MI355_ERR_INVALID, "argmax_chunked: null pointer");
  MI355_REQUIRE(rows >= 0 && rows <= 65535 && vocab > 0 && vocab < (1LL << 31) - 1 && row_stride >= vocab, MI355_ERR_INVALID,
                "argmax_chunked: bad shape");
  if (rows == 0) return MI355_OK;
  // enough workgroups for every CU; a chunk keeps at least 16 K elements
  int64_t chunks = (256 + rows - 1) / rows;
  chunks = std::min<int64_t>(chunks, std::max<int64_t>(1, vocab / 16384));
  chunks = std::min<int64_t>(chunks, 16);
  if (chunks <= 1 || !workspace) {
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)rows), dim3(SAMP_THREADS), 0, (hipStream_t)stream, ids, logits, vocab,
                       row_stride);
  } else {
    MI355_REQUIRE(workspace_bytes >= mi355_argmax_workspace_bytes(rows) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                  MI355_ERR_INVALID, "argmax_chunked: workspace of %lld bytes (8-byte aligned, zeroed) required",
                  (long long)mi355_argmax_workspace_bytes(rows));
    unsigned long long* cells = static_cast<unsigned long long*>(workspace);
    unsigned* tickets = reinterpret_cast<unsigned*>(cells + rows);
    hipLaunchKernelGGL(argmax_chunk_kernel, dim3((unsigned)rows, (unsigned)chunks), dim3(SAMP_THREADS), 0,
                       (hipStream_t)stream, ids, logits, vocab, row_stride, cells, tickets);
  }
  MI355_CHECK_LAUNCH("argmax_chunked");
  return MI355_OK;
}

int mi355_softmax_temperature(float* logits, const float* temperatures, int64_t rows,
                              int64_t vocab, int64_t row_stride, void* stream) {
  MI355_REQUIRE(logits && temperatures, MI355_ERR_INVALID, "softmax_temperature: null pointer");
  MI355_REQUIRE(rows >= 0 && vocab > 0 && row_stride >= vocab, MI355_ERR_INVALID,
                "softmax_temperature: bad shape");
  if (rows == 0) return MI355_OK;
  hipLaunchKernelGGL(softmax_temperature_kernel, dim3((unsigned)rows), dim3(SAMP_THREADS), 0,
                     (hipStream_t)stream, logits, temperatures, vocab, row_stride);
  MI355_CHECK_LAUNCH("softmax_temperature");
  return MI355_OK;
}

int mi355_top_k_top_p_min_p_sample(int32_t* ids, const float* probs, const int32_t* top_ks,
                                   const float* top_ps, const float* min_ps, const float* uniform,
                                   int64_t rows, int64_t vocab, int64_t row_stride, void* stream) {
  MI355_REQUIRE(ids && probs && uniform, MI355_ERR_INVALID, "top_k_top_p_min_p_sample: null pointer");
  MI355_REQUIRE(rows >= 0 && vocab > 0 && vocab < (1LL << 31) && row_stride >= vocab,
                MI355_ERR_INVALID, "top_k_top_p_min_p_sample: bad shape");
  if (rows == 0) return MI355_OK;
  hipLaunchKernelGGL(topk_topp_minp_sample_kernel, dim3((unsigned)rows), dim3(SAMP_THREADS), 0,
                     (hipStream_t)stream, ids, probs, top_ks, top_ps, min_ps, uniform, vocab,
                     row_stride);
  MI355_CHECK_LAUNCH("top_k_top_p_min_p_sample");
  return MI355_OK;
}

}  // extern "C"
