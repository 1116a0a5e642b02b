// Sum all-reduce across the GPUs of one xGMI hive for decode-sized tensors (a few KiB to a few
// MiB), where a ring collective is bound by its 2(n-1) hops of latency rather than by bandwidth.
//
// Every rank owns one shared region (uncached device memory, opened by its peers through IPC):
//
//     [ ArSignal | staging: max_bytes | scatter: max_bytes ]
//
// One launch does the whole collective.  A workgroup first copies its share of the rank's input
// into the rank's own staging area, then meets the same-numbered workgroup of every peer at a flag
// barrier, then pulls that share from every peer's staging area over xGMI (every link is busy at
// once: xGMI is point to point) and sums it in fp32 in rank order 0..n-1, rounding once.  All ranks
// therefore produce the same bits, and both algorithms below produce the same bits as each other:
//
//   one-shot   every rank sums the whole tensor: (n-1) * bytes pulled per rank, two barriers
//   two-shot   rank r sums slice r into its scatter area, a barrier, then every rank gathers the n
//              reduced slices: 2 (n-1)/n * bytes pulled per rank, three barriers
//
// Barriers are per workgroup: a workgroup touches only indices that the same-numbered workgroup of
// a peer wrote (the index sets are built to make that true), so workgroup b never waits for any
// other workgroup.  Flags are 32-bit counters that only grow (compared through their signed
// difference, so wrap-around is harmless); they are written with vector atomic stores at system
// scope.  A wait gives up after AR_TIMEOUT_TICKS of the 100 MHz wall clock and counts the event in
// the communicator's status word: a lost peer ends in an error code, not in a hung GPU.
//
// Follows what the reference's kernels do (sgl-kernel/csrc/allreduce/custom_all_reduce_hip.cuh:
// 151-248 barriers, 260-282 one stage, 293-343 two stages) with three differences: the copy of the
// input into IPC-visible memory is part of the kernel (the reference issues a separate device copy
// outside graphs and registers graph buffers inside them, custom_all_reduce.hip:97-110), the
// two-stage reduction keeps rank order instead of rotating it, and waits are bounded.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>

#include "common.cuh"

namespace mi355 {

constexpr int AR_MAX_RANKS = 8;
constexpr int AR_MAX_BLOCKS = 64;
constexpr int AR_THREADS = 512;
constexpr long long AR_TICKS_PER_MS = 100000LL;       // the constant clock runs at 100 MHz
constexpr long long AR_DEFAULT_TIMEOUT_MS = 20000LL;  // far above any rank skew a healthy job shows

struct ArSignal {
  // sig[b][r]: written by rank r's workgroup b, read by the owner's workgroup b
  alignas(128) uint32_t sig[AR_MAX_BLOCKS][AR_MAX_RANKS];
  // epoch[b]: how many barriers the owner's workgroup b has passed (only the owner touches it)
  alignas(128) uint32_t epoch[AR_MAX_BLOCKS];
};
static_assert(sizeof(ArSignal) % 256 == 0, "staging area must stay 256-byte aligned");

struct ArPeers {
  char* base[AR_MAX_RANKS];   // every rank's region as mapped into this process
};

struct ArComm {
  int rank, world;
  int64_t max_bytes;
  ArPeers peers;
  uint32_t* status;           // device word: number of waits that timed out
  long long timeout_ticks;    // how long a barrier wait may last before the call gives up
  hipStream_t poll_stream;    // non-blocking stream for reading `status` beside running work
};

__device__ __forceinline__ ArSignal* ar_signal(char* base) { return reinterpret_cast<ArSignal*>(base); }
__device__ __forceinline__ char* ar_staging(char* base) { return base + sizeof(ArSignal); }
__device__ __forceinline__ char* ar_scatter(char* base, int64_t max_bytes) { return base + sizeof(ArSignal) + max_bytes; }

// Meet workgroup blockIdx.x of every peer.  Everything this workgroup stored before the call is
// visible to the peers' workgroups after it, and the other way round.
// `epoch` is the workgroup's barrier count, loaded once per launch and stored back at its end.
template <int NR>
__device__ __forceinline__ bool ar_barrier(const ArPeers& p, int rank, uint32_t& epoch, uint32_t* status,
                                           long long timeout_ticks) {
  // Every wave completes its own stores and meets the others; the first wave then publishes them
  // for the whole workgroup with ONE system-scope release (a write-back of the cache level that
  // other devices do not see) instead of one per wave, and takes ONE system-scope acquire after
  // the last flag has arrived instead of one per poll.
  // Returns false (to every thread of the workgroup) when a peer did not arrive in time: the wait
  // is counted in *status and the caller must not publish what it computes from peer data.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  ArSignal* self = ar_signal(p.base[rank]);
  const uint32_t want = ++epoch;
  int late = 0;
  if (threadIdx.x < NR) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __scoped_atomic_store_n(&ar_signal(p.base[threadIdx.x])->sig[blockIdx.x][rank], want, __ATOMIC_RELAXED,
                            __MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    while ((int32_t)(__scoped_atomic_load_n(&self->sig[blockIdx.x][threadIdx.x], __ATOMIC_RELAXED,
                                            __MEMORY_SCOPE_SYSTEM) - want) < 0) {
      if (wall_clock64() - t0 > timeout_ticks) {
        atomicAdd(status, 1u);
        late = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  late = __syncthreads_or(late);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return late == 0;
}

__device__ __forceinline__ uint32_t ar_load_epoch(const ArPeers& p, int rank) {
  return ar_signal(p.base[rank])->epoch[blockIdx.x];
}
__device__ __forceinline__ void ar_store_epoch(const ArPeers& p, int rank, uint32_t epoch) {
  if (threadIdx.x == 0) ar_signal(p.base[rank])->epoch[blockIdx.x] = epoch;
}

// 16 bytes: 8 16-bit values or 4 floats.  A plain aggregate (not HIP's uint4 class): arrays of it
// are split into registers by the compiler at every rank count.
struct alignas(16) ArVec { uint32_t x, y, z, w; };

// What a call leaves in its output when a barrier timed out: every 16-bit lane 0x7fc0, which reads
// as NaN in bf16, f16 and (pairwise) f32 -- a wrong sum must never look like a number.
__device__ __forceinline__ ArVec ar_poison() { return ArVec{0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}; }

template <typename T> struct ArAcc {
  static constexpr int N = 16 / sizeof(T);
  float v[N];
  __device__ __forceinline__ void set(const ArVec& raw) {
    const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = to_f32(e[i]);
  }
  __device__ __forceinline__ void add(const ArVec& raw) {
    const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += to_f32(e[i]);
  }
  __device__ __forceinline__ ArVec get() const {
    ArVec raw;
    T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = from_f32<T>(v[i]);
    return raw;
  }
};

template <typename T, int NR>
__device__ __forceinline__ ArVec ar_reduce(const ArVec* const* src, int64_t idx) {
  ArVec raw[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) raw[r] = src[r][idx];   // all pulls in flight together
  ArAcc<T> acc;
  acc.set(raw[0]);
#pragma unroll
  for (int r = 1; r < NR; ++r) acc.add(raw[r]);
  return acc.get();
}

template <typename T, int NR>
__global__ __launch_bounds__(AR_THREADS) void ar_one_shot_kernel(ArPeers p, const ArVec* in, ArVec* out,
                                                                 int64_t nvec, int rank, uint32_t* status,
                                                                 long long timeout) {
  const int64_t tid = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * AR_THREADS;
  ArVec* mine = reinterpret_cast<ArVec*>(ar_staging(p.base[rank]));
  uint32_t epoch = ar_load_epoch(p, rank);
  for (int64_t i = tid; i < nvec; i += stride) mine[i] = in[i];
  bool ok = ar_barrier<NR>(p, rank, epoch, status, timeout);
  const ArVec* src[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) src[r] = reinterpret_cast<const ArVec*>(ar_staging(p.base[r]));
  for (int64_t i = tid; i < nvec; i += stride) out[i] = ok ? ar_reduce<T, NR>(src, i) : ar_poison();
  // nobody may overwrite its staging area (the next call) while a peer still reads it
  ar_barrier<NR>(p, rank, epoch, status, timeout);
  ar_store_epoch(p, rank, epoch);
}

template <int NR>
__device__ __forceinline__ void gather_slices(const ArVec* const* red, ArVec* __restrict__ out, int64_t tid,
                                              int64_t stride, int64_t part, int64_t last);

template <typename T, int NR>
__global__ __launch_bounds__(AR_THREADS) void ar_two_shot_kernel(ArPeers p, const ArVec* in, ArVec* out,
                                                                 int64_t nvec, int rank, int64_t max_bytes,
                                                                 uint32_t* status, long long timeout) {
  const int64_t tid = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * AR_THREADS;
  const int64_t part = nvec / NR;                       // slice r = [r*part, (r+1)*part), the last one
  const int64_t last = nvec - part * (NR - 1);          // takes the remainder
  ArVec* mine = reinterpret_cast<ArVec*>(ar_staging(p.base[rank]));
  uint32_t epoch = ar_load_epoch(p, rank);
  // copy-in, slice by slice: thread t of workgroup b writes, in every slice, exactly the offsets
  // that thread t of workgroup b of the slice's owner reads below
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int64_t len = r == NR - 1 ? last : part;
    for (int64_t i = tid; i < len; i += stride) mine[r * part + i] = in[r * part + i];
  }
  bool ok = ar_barrier<NR>(p, rank, epoch, status, timeout);
  const ArVec* src[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) src[r] = reinterpret_cast<const ArVec*>(ar_staging(p.base[r])) + rank * part;
  ArVec* reduced = reinterpret_cast<ArVec*>(ar_scatter(p.base[rank], max_bytes));
  const int64_t my_len = rank == NR - 1 ? last : part;
  // a slice reduced from data that may not have arrived is published as NaN, so the peers that
  // gather it see the failure too
  for (int64_t i = tid; i < my_len; i += stride) reduced[i] = ok ? ar_reduce<T, NR>(src, i) : ar_poison();
  ok = ar_barrier<NR>(p, rank, epoch, status, timeout) && ok;
  const ArVec* red[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) red[r] = reinterpret_cast<const ArVec*>(ar_scatter(p.base[r], max_bytes));
  gather_slices<NR>(red, out, tid, stride, part, last);
  if (!ok)
    for (int64_t i = tid; i < nvec; i += stride) out[i] = ar_poison();
  ar_barrier<NR>(p, rank, epoch, status, timeout);
  ar_store_epoch(p, rank, epoch);
}

// Stage 2 of the two-shot kernel.  `out` never overlaps a scatter area; saying so lets the compiler
// keep all loads of an index in registers ahead of its stores.
template <int NR>
__device__ __forceinline__ void gather_slices(const ArVec* const* red, ArVec* __restrict__ out, int64_t tid,
                                              int64_t stride, int64_t part, int64_t last) {
  for (int64_t i = tid; i < last; i += stride) {
    // loads are unconditional (index clamped into the slice) so that all of them are in flight together
    const int64_t ic = i < part ? i : (part > 0 ? part - 1 : 0);
    constexpr int H = NR / 2;   // (two half-size register arrays, see the all-gather below)
    ArVec lo[H], hi[NR - H];
#pragma unroll
    for (int r = 0; r < H; ++r) lo[r] = red[r][ic];
#pragma unroll
    for (int r = H; r < NR; ++r) hi[r - H] = red[r][r == NR - 1 ? i : ic];
#pragma unroll
    for (int r = 0; r < H; ++r)
      if (i < part) out[r * part + i] = lo[r];
#pragma unroll
    for (int r = H; r < NR; ++r)
      if (r == NR - 1 || i < part) out[r * part + i] = hi[r - H];
  }
}

// out[r] = rank r's input, r = 0..n-1: the concatenating all-gather of vocab-sharded logits, with
// the same copy-in / barrier / pull structure (no arithmetic, so no element type).
template <int NR>
__global__ __launch_bounds__(AR_THREADS) void ar_all_gather_kernel(ArPeers p, const ArVec* __restrict__ in, ArVec* __restrict__ out, int64_t nvec,
                                                                   int rank, uint32_t* status, long long timeout) {
  const int64_t tid = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * AR_THREADS;
  ArVec* mine = reinterpret_cast<ArVec*>(ar_staging(p.base[rank]));
  uint32_t epoch = ar_load_epoch(p, rank);
  for (int64_t i = tid; i < nvec; i += stride) mine[i] = in[i];
  const bool ok = ar_barrier<NR>(p, rank, epoch, status, timeout);
  const ArVec* src[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) src[r] = reinterpret_cast<const ArVec*>(ar_staging(p.base[r]));
  for (int64_t i = tid; i < nvec; i += stride) {
    // two half-size register arrays: the compiler keeps arrays of up to four vectors in registers
    // and sends longer ones through private memory
    constexpr int H = NR / 2;
    ArVec lo[H], hi[NR - H];
#pragma unroll
    for (int r = 0; r < H; ++r) lo[r] = src[r][i];
#pragma unroll
    for (int r = H; r < NR; ++r) hi[r - H] = src[r][i];
#pragma unroll
    for (int r = 0; r < H; ++r) out[r * nvec + i] = lo[r];
#pragma unroll
    for (int r = H; r < NR; ++r) out[r * nvec + i] = hi[r - H];
  }
  if (!ok)
    for (int64_t i = tid; i < nvec * NR; i += stride) out[i] = ar_poison();
  ar_barrier<NR>(p, rank, epoch, status, timeout);
  ar_store_epoch(p, rank, epoch);
}

// All-reduce + residual add + RMSNorm in one launch, for the row-parallel projections of a tensor-
// parallel decoder layer: x (this rank's partial sums) -> sum over ranks, rounded to T exactly as
// the plain all-reduce rounds it -> + residual (kept in fp32 for the norm, stored rounded) ->
// RMSNorm * weight.  The arithmetic after the sum repeats the fused add+RMSNorm kernel's statement
// by statement (same lanes per row, same per-lane accumulation and block reduction), so the result equals
// all-reduce followed by that kernel bit for bit.  A workgroup owns whole rows: b, b + grid, ...
// replaces: RMSNorm.forward_with_allreduce_fusion (layers/layernorm.py:166-190) and the deferred
//           all-reduce of LayerCommunicator.prepare_mlp / prepare_attn (layers/communicator.py).
template <typename T, int NR, int CHUNKS, int BT>
__global__ __launch_bounds__(BT) void ar_fused_add_rmsnorm_kernel(ArPeers p, T* x, T* residual, const T* __restrict__ w,
                                                                   int rows, int64_t hidden, float eps, int rank,
                                                                   uint32_t* status, long long timeout) {
  constexpr int VE = 16 / sizeof(T);
  __shared__ float red[16];
  const int nvec = (int)(hidden / VE);
  ArVec* mine = reinterpret_cast<ArVec*>(ar_staging(p.base[rank]));
  uint32_t epoch = ar_load_epoch(p, rank);
  Vec<T, VE> wv[CHUNKS];
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int vi = threadIdx.x + c * BT;
    wv[c] = ldvec<T, VE>(w + (int64_t)(vi < nvec ? vi : 0) * VE);
  }
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const ArVec* xin = reinterpret_cast<const ArVec*>(x + (int64_t)row * hidden);
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      const int vi = threadIdx.x + c * BT;
      if (vi < nvec) mine[(int64_t)row * nvec + vi] = xin[vi];
    }
  }
  const bool ok = ar_barrier<NR>(p, rank, epoch, status, timeout);
  const ArVec* src[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) src[r] = reinterpret_cast<const ArVec*>(ar_staging(p.base[r]));
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    T* xr = x + (int64_t)row * hidden;
    T* rr = residual + (int64_t)row * hidden;
    Vec<T, VE> a[CHUNKS], b[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      const int vi = threadIdx.x + c * BT;
      const int64_t v0 = vi < nvec ? vi : 0;
      b[c] = ldvec<T, VE>(rr + v0 * VE);
      const ArVec summed = ar_reduce<T, NR>(src, (int64_t)row * nvec + v0);
      a[c] = *reinterpret_cast<const Vec<T, VE>*>(&summed);
    }
    float vals[CHUNKS][VE];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      const int vi = threadIdx.x + c * BT;
      if (vi < nvec) {
        Vec<T, VE> s;
#pragma unroll
        for (int j = 0; j < VE; ++j) {
          float f = to_f32(a[c].v[j]) + to_f32(b[c].v[j]);
          vals[c][j] = f;
          s.v[j] = from_f32<T>(f);
        }
        stvec<T, VE>(rr + (int64_t)vi * VE, s);
#pragma unroll
        for (int j = 0; j < VE; ++j) ss += vals[c][j] * vals[c][j];
      }
    }
    ss = block_sum(ss, red);
    const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      const int vi = threadIdx.x + c * BT;
      if (vi < nvec) {
        Vec<T, VE> ov;
#pragma unroll
        for (int j = 0; j < VE; ++j) ov.v[j] = from_f32<T>((vals[c][j] * inv) * to_f32(wv[c].v[j]));
        stvec<T, VE>(xr + (int64_t)vi * VE, ov);
      }
    }
  }
  if (!ok) {   // a peer's partial sums never arrived: both outputs become NaN
    for (int row = blockIdx.x; row < rows; row += gridDim.x)
      for (int vi = threadIdx.x; vi < nvec; vi += BT) {
        reinterpret_cast<ArVec*>(x + (int64_t)row * hidden)[vi] = ar_poison();
        reinterpret_cast<ArVec*>(residual + (int64_t)row * hidden)[vi] = ar_poison();
      }
  }
  ar_barrier<NR>(p, rank, epoch, status, timeout);
  ar_store_epoch(p, rank, epoch);
}

template <typename T, int NR>
static void ar_launch(const ArComm* c, void* out, const void* in, int64_t nvec, int algo, hipStream_t st) {
  const int grid = (int)std::min<int64_t>(AR_MAX_BLOCKS, std::max<int64_t>(1, (nvec + AR_THREADS - 1) / AR_THREADS));
  if (algo == 1)
    ar_one_shot_kernel<T, NR><<<grid, AR_THREADS, 0, st>>>(c->peers, (const ArVec*)in, (ArVec*)out, nvec, c->rank, c->status,
                                                          c->timeout_ticks);
  else
    ar_two_shot_kernel<T, NR><<<grid, AR_THREADS, 0, st>>>(c->peers, (const ArVec*)in, (ArVec*)out, nvec, c->rank,
                                                          c->max_bytes, c->status, c->timeout_ticks);
}

template <typename T>
static int ar_dispatch_world(const ArComm* c, void* out, const void* in, int64_t nvec, int algo, hipStream_t st) {
  switch (c->world) {
    case 2: ar_launch<T, 2>(c, out, in, nvec, algo, st); break;
    case 4: ar_launch<T, 4>(c, out, in, nvec, algo, st); break;
    case 6: ar_launch<T, 6>(c, out, in, nvec, algo, st); break;
    case 8: ar_launch<T, 8>(c, out, in, nvec, algo, st); break;
    default: set_error("mi355_ar_all_reduce: world size %d (2, 4, 6 or 8)", c->world); return MI355_ERR_UNSUPPORTED;
  }
  return MI355_OK;
}

template <typename T, int NR>
static int ar_fused_norm_launch(const ArComm* c, T* x, T* residual, const T* w, int rows, int64_t hidden, float eps,
                                hipStream_t st) {
  const int nvec = (int)(hidden / (16 / sizeof(T)));
  const int chunks = (nvec + 255) / 256;
  const int grid = std::min(AR_MAX_BLOCKS, rows);
  // lanes per row and chunks per lane exactly as the fused add+RMSNorm launch picks them (csrc/elementwise.hip)
#define AR_NORM(C, BT_) \
  ar_fused_add_rmsnorm_kernel<T, NR, C, BT_><<<grid, BT_, 0, st>>>(c->peers, x, residual, w, rows, hidden, eps, c->rank, c->status, \
                                                                   c->timeout_ticks)
  if (chunks <= 1) AR_NORM(1, 256);
  else if (chunks <= 2) AR_NORM(1, 512);
  else if (chunks <= 4) AR_NORM(2, 512);
  else AR_NORM(4, 512);
#undef AR_NORM
  return MI355_OK;
}

template <typename T>
static int ar_fused_norm_world(const ArComm* c, void* x, void* residual, const void* w, int rows, int64_t hidden,
                               float eps, hipStream_t st) {
  switch (c->world) {
    case 2: return ar_fused_norm_launch<T, 2>(c, (T*)x, (T*)residual, (const T*)w, rows, hidden, eps, st);
    case 4: return ar_fused_norm_launch<T, 4>(c, (T*)x, (T*)residual, (const T*)w, rows, hidden, eps, st);
    case 6: return ar_fused_norm_launch<T, 6>(c, (T*)x, (T*)residual, (const T*)w, rows, hidden, eps, st);
    case 8: return ar_fused_norm_launch<T, 8>(c, (T*)x, (T*)residual, (const T*)w, rows, hidden, eps, st);
    default: set_error("mi355_ar_fused_add_rmsnorm: world size %d (2, 4, 6 or 8)", c->world); return MI355_ERR_UNSUPPORTED;
  }
}

}  // namespace mi355

using namespace mi355;

#define AR_HIP(call, name)                                                 \
  do {                                                                     \
    hipError_t e__ = (call);                                               \
    if (e__ != hipSuccess) {                                               \
      set_error("%s: %s: %s", name, #call, hipGetErrorString(e__));        \
      return MI355_ERR_LAUNCH;                                             \
    }                                                                      \
  } while (0)

extern "C" {

int64_t mi355_ar_region_bytes(int64_t max_bytes) {
  if (max_bytes <= 0 || max_bytes % 256 != 0) return -1;
  return (int64_t)sizeof(ArSignal) + 2 * max_bytes;
}

int mi355_ar_alloc_region(void** region, int64_t max_bytes) {
  MI355_REQUIRE(region && mi355_ar_region_bytes(max_bytes) > 0, MI355_ERR_INVALID,
                "mi355_ar_alloc_region: max_bytes must be a positive multiple of 256");
  const size_t bytes = (size_t)mi355_ar_region_bytes(max_bytes);
  // uncached: flags and payload written by one GPU (or one XCD) must be seen by the others without
  // waiting for a kernel boundary
  AR_HIP(hipExtMallocWithFlags(region, bytes, hipDeviceMallocUncached), "mi355_ar_alloc_region");
  AR_HIP(hipMemset(*region, 0, bytes), "mi355_ar_alloc_region");
  AR_HIP(hipDeviceSynchronize(), "mi355_ar_alloc_region");
  return MI355_OK;
}

int mi355_ar_free_region(void* region) {
  AR_HIP(hipFree(region), "mi355_ar_free_region");
  return MI355_OK;
}

int mi355_ar_ipc_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int mi355_ar_get_ipc_handle(const void* region, uint8_t* handle) {
  MI355_REQUIRE(region && handle, MI355_ERR_INVALID, "mi355_ar_get_ipc_handle: null argument");
  hipIpcMemHandle_t h;
  AR_HIP(hipIpcGetMemHandle(&h, const_cast<void*>(region)), "mi355_ar_get_ipc_handle");
  std::memcpy(handle, &h, sizeof(h));
  return MI355_OK;
}

int mi355_ar_open_ipc_handle(const uint8_t* handle, void** region) {
  MI355_REQUIRE(region && handle, MI355_ERR_INVALID, "mi355_ar_open_ipc_handle: null argument");
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  AR_HIP(hipIpcOpenMemHandle(region, h, hipIpcMemLazyEnablePeerAccess), "mi355_ar_open_ipc_handle");
  return MI355_OK;
}

int mi355_ar_close_ipc_handle(void* region) {
  AR_HIP(hipIpcCloseMemHandle(region), "mi355_ar_close_ipc_handle");
  return MI355_OK;
}

int mi355_ar_init(void** comm, void* const* regions, int rank, int world, int64_t max_bytes) {
  MI355_REQUIRE(comm && regions, MI355_ERR_INVALID, "mi355_ar_init: null argument");
  MI355_REQUIRE(world == 2 || world == 4 || world == 6 || world == 8, MI355_ERR_UNSUPPORTED,
                "mi355_ar_init: world size %d (2, 4, 6 or 8)", world);
  MI355_REQUIRE(rank >= 0 && rank < world, MI355_ERR_INVALID, "mi355_ar_init: rank %d of %d", rank, world);
  MI355_REQUIRE(mi355_ar_region_bytes(max_bytes) > 0, MI355_ERR_INVALID,
                "mi355_ar_init: max_bytes must be a positive multiple of 256");
  ArComm* c = new (std::nothrow) ArComm();
  MI355_REQUIRE(c, MI355_ERR_INVALID, "mi355_ar_init: out of host memory");
  c->rank = rank;
  c->world = world;
  c->max_bytes = max_bytes;
  c->timeout_ticks = AR_DEFAULT_TIMEOUT_MS * AR_TICKS_PER_MS;
  for (int r = 0; r < AR_MAX_RANKS; ++r) c->peers.base[r] = nullptr;
  for (int r = 0; r < world; ++r) {
    if (!regions[r] || !aligned16(regions[r])) {
      delete c;
      set_error("mi355_ar_init: region of rank %d is null or misaligned", r);
      return MI355_ERR_INVALID;
    }
    c->peers.base[r] = static_cast<char*>(regions[r]);
  }
  if (hipMalloc(&c->status, 256) != hipSuccess || hipMemset(c->status, 0, 256) != hipSuccess ||
      hipStreamCreateWithFlags(&c->poll_stream, hipStreamNonBlocking) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    delete c;
    set_error("mi355_ar_init: could not allocate the status word");
    return MI355_ERR_LAUNCH;
  }
  *comm = c;
  return MI355_OK;
}

int mi355_ar_set_timeout_ms(void* comm, int64_t ms) {
  MI355_REQUIRE(comm && ms > 0 && ms <= 3600000, MI355_ERR_INVALID, "mi355_ar_set_timeout_ms: null communicator or %lld ms outside 1..3600000",
                (long long)ms);
  static_cast<ArComm*>(comm)->timeout_ticks = ms * AR_TICKS_PER_MS;
  return MI355_OK;
}

int mi355_ar_dispose(void* comm) {
  if (!comm) return MI355_OK;
  ArComm* c = static_cast<ArComm*>(comm);
  (void)hipFree(c->status);
  (void)hipStreamDestroy(c->poll_stream);
  delete c;
  return MI355_OK;
}

int mi355_ar_choose_algo(int world, int64_t bytes) {
  // one-shot pulls (n-1) * bytes over n-1 links and saves a barrier; two-shot pulls 2 (n-1)/n of
  // them.  With two ranks the volumes are equal; from 4 ranks on the extra barrier (one xGMI round
  // trip, ~2 us) is repaid once a link carries more than ~128 KiB.
  if (world <= 2) return 1;
  // Same switch points as the reference (custom_all_reduce_hip.cuh:543-550).
  return bytes < (world <= 4 ? 512 : 256) * 1024 ? 1 : 2;
}

int mi355_ar_all_reduce(void* comm, void* out, const void* in, int64_t count, int dtype, int algo, void* stream) {
  MI355_REQUIRE(comm && out && in, MI355_ERR_INVALID, "mi355_ar_all_reduce: null argument");
  const ArComm* c = static_cast<const ArComm*>(comm);
  const int64_t esize = dtype == MI355_F32 ? 4 : 2;
  const int64_t bytes = count * esize;
  MI355_REQUIRE(count > 0 && bytes % 16 == 0, MI355_ERR_INVALID,
                "mi355_ar_all_reduce: %lld bytes is not a positive multiple of 16", (long long)bytes);
  MI355_REQUIRE(bytes <= c->max_bytes, MI355_ERR_INVALID, "mi355_ar_all_reduce: %lld bytes exceed the region's %lld",
                (long long)bytes, (long long)c->max_bytes);
  MI355_REQUIRE(aligned16(in) && aligned16(out), MI355_ERR_INVALID, "mi355_ar_all_reduce: buffers must be 16-byte aligned");
  MI355_REQUIRE(algo >= 0 && algo <= 2, MI355_ERR_INVALID, "mi355_ar_all_reduce: algo %d (0 auto, 1 one-shot, 2 two-shot)", algo);
  if (algo == 0) algo = mi355_ar_choose_algo(c->world, bytes);
  const int64_t nvec = bytes / 16;
  if (algo == 2 && nvec < c->world) algo = 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = MI355_OK;
  MI355_DISPATCH_DTYPE(dtype, "mi355_ar_all_reduce", rc = ar_dispatch_world<T>(c, out, in, nvec, algo, st));
  if (rc != MI355_OK) return rc;
  MI355_CHECK_LAUNCH("mi355_ar_all_reduce");
  return MI355_OK;
}

int mi355_ar_all_gather(void* comm, void* out, const void* in, int64_t bytes, void* stream) {
  MI355_REQUIRE(comm && out && in, MI355_ERR_INVALID, "mi355_ar_all_gather: null argument");
  const ArComm* c = static_cast<const ArComm*>(comm);
  MI355_REQUIRE(bytes > 0 && bytes % 16 == 0 && bytes <= c->max_bytes, MI355_ERR_INVALID,
                "mi355_ar_all_gather: %lld bytes per rank (a positive multiple of 16, at most %lld)", (long long)bytes,
                (long long)c->max_bytes);
  MI355_REQUIRE(aligned16(in) && aligned16(out), MI355_ERR_INVALID, "mi355_ar_all_gather: buffers must be 16-byte aligned");
  MI355_REQUIRE(in != out, MI355_ERR_INVALID, "mi355_ar_all_gather: in place is not supported");
  const int64_t nvec = bytes / 16;
  const int grid = (int)std::min<int64_t>(AR_MAX_BLOCKS, (nvec + AR_THREADS - 1) / AR_THREADS);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define AR_GATHER(NR) \
  case NR: ar_all_gather_kernel<NR><<<grid, AR_THREADS, 0, st>>>(c->peers, (const ArVec*)in, (ArVec*)out, nvec, c->rank, c->status, c->timeout_ticks); break
  switch (c->world) {
    AR_GATHER(2);
    AR_GATHER(4);
    AR_GATHER(6);
    AR_GATHER(8);
    default: set_error("mi355_ar_all_gather: world size %d (2, 4, 6 or 8)", c->world); return MI355_ERR_UNSUPPORTED;
  }
#undef AR_GATHER
  MI355_CHECK_LAUNCH("mi355_ar_all_gather");
  return MI355_OK;
}

int mi355_ar_fused_add_rmsnorm(void* comm, void* x, void* residual, const void* weight, int64_t rows, int64_t hidden,
                               float eps, int dtype, void* stream) {
  MI355_REQUIRE(comm && x && residual && weight, MI355_ERR_INVALID, "mi355_ar_fused_add_rmsnorm: null argument");
  const ArComm* c = static_cast<const ArComm*>(comm);
  const int64_t esize = dtype == MI355_F32 ? 4 : 2;
  const int64_t ve = 16 / esize;
  MI355_REQUIRE(rows > 0 && rows <= (1 << 24) && hidden > 0 && hidden % ve == 0 && hidden / ve <= 256 * 8,
                MI355_ERR_UNSUPPORTED, "mi355_ar_fused_add_rmsnorm: rows %lld, hidden %lld (hidden a multiple of %lld, at most %lld)",
                (long long)rows, (long long)hidden, (long long)ve, (long long)(256 * 8 * ve));
  MI355_REQUIRE(rows * hidden * esize <= c->max_bytes, MI355_ERR_INVALID,
                "mi355_ar_fused_add_rmsnorm: %lld bytes exceed the region's %lld", (long long)(rows * hidden * esize),
                (long long)c->max_bytes);
  MI355_REQUIRE(aligned16(x) && aligned16(residual) && aligned16(weight), MI355_ERR_INVALID,
                "mi355_ar_fused_add_rmsnorm: buffers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = MI355_OK;
  MI355_DISPATCH_DTYPE(dtype, "mi355_ar_fused_add_rmsnorm",
                       rc = ar_fused_norm_world<T>(c, x, residual, weight, (int)rows, hidden, eps, st));
  if (rc != MI355_OK) return rc;
  MI355_CHECK_LAUNCH("mi355_ar_fused_add_rmsnorm");
  return MI355_OK;
}

int mi355_ar_poll_status(void* comm) {
  MI355_REQUIRE(comm, MI355_ERR_INVALID, "mi355_ar_poll_status: null communicator");
  const ArComm* c = static_cast<const ArComm*>(comm);
  uint32_t n = 0;
  AR_HIP(hipMemcpyAsync(&n, c->status, sizeof(n), hipMemcpyDeviceToHost, c->poll_stream), "mi355_ar_poll_status");
  AR_HIP(hipStreamSynchronize(c->poll_stream), "mi355_ar_poll_status");
  if (n) {
    set_error("mi355_ar_poll_status: %u barrier waits timed out (a peer did not arrive)", n);
    return MI355_ERR_LAUNCH;
  }
  return MI355_OK;
}

int mi355_ar_status(void* comm, void* stream) {
  MI355_REQUIRE(comm, MI355_ERR_INVALID, "mi355_ar_status: null communicator");
  const ArComm* c = static_cast<const ArComm*>(comm);
  uint32_t n = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  AR_HIP(hipStreamSynchronize(st), "mi355_ar_status");
  AR_HIP(hipMemcpy(&n, c->status, sizeof(n), hipMemcpyDeviceToHost), "mi355_ar_status");
  if (n) {
    set_error("mi355_ar_status: %u barrier waits timed out (a peer did not arrive)", n);
    return MI355_ERR_LAUNCH;
  }
  return MI355_OK;
}

}  // extern "C"
