// Integer bookkeeping kernels of the forward-batch path: page-table gathers and writes.
// All are exact integer work; one workgroup per request, lanes stride the token axis so the
// req_to_token row and the flat output are read/written in coalesced 256-B segments.
#include "common.cuh"

namespace mi355 {

template <typename PT, typename LT>
__global__ __launch_bounds__(256) void create_kv_indices_kernel(
    const int32_t* __restrict__ req_to_token, const PT* __restrict__ req_pool_indices,
    const LT* __restrict__ page_kernel_lens, const int32_t* __restrict__ kv_indptr,
    const int32_t* __restrict__ kv_start_idx, int32_t* __restrict__ kv_indices, int64_t stride) {
  const int b = blockIdx.x;
  const int64_t pool = (int64_t)req_pool_indices[b];
  const int64_t out0 = kv_indptr[b];
  const int64_t start = kv_start_idx ? kv_start_idx[b] : 0;
  const int64_t len = (int64_t)page_kernel_lens[b];
  const int32_t* src = req_to_token + pool * stride + start;
  for (int64_t i = threadIdx.x; i < len; i += 256) kv_indices[out0 + i] = src[i];
}

__global__ __launch_bounds__(256) void write_req_to_token_kernel(
    int32_t* __restrict__ req_to_token, const int64_t* __restrict__ req_pool_indices,
    const int64_t* __restrict__ pre_lens, const int64_t* __restrict__ seq_lens,
    const int64_t* __restrict__ extend_lens, const int64_t* __restrict__ out_cache_loc,
    int64_t stride) {
  const int b = blockIdx.x;
  // exclusive prefix sum of extend_lens[0..b): wave-parallel instead of the serial loop
  __shared__ int64_t red[4];
  int64_t part = 0;
  for (int i = threadIdx.x; i < b; i += 256) part += extend_lens[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  const int64_t start = red[0] + red[1] + red[2] + red[3];
  const int64_t pre = pre_lens[b], n = seq_lens[b] - pre;
  int32_t* dst = req_to_token + req_pool_indices[b] * stride + pre;
  for (int64_t i = threadIdx.x; i < n; i += 256) dst[i] = (int32_t)out_cache_loc[start + i];
}

__global__ __launch_bounds__(256) void compute_position_kernel(
    const int32_t* __restrict__ prefix_lens, const int32_t* __restrict__ seq_lens,
    int64_t* __restrict__ positions, int32_t* __restrict__ extend_start_loc) {
  const int b = blockIdx.x;
  __shared__ int64_t red[4];
  int64_t part = 0;
  for (int i = threadIdx.x; i < b; i += 256) part += seq_lens[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  const int64_t start = red[0] + red[1] + red[2] + red[3];
  const int64_t pre = prefix_lens ? prefix_lens[b] : 0, n = seq_lens[b];
  for (int64_t i = threadIdx.x; i < n; i += 256) positions[start + i] = pre + i;
  if (threadIdx.x == 0) extend_start_loc[b] = (int32_t)start;
}

__global__ __launch_bounds__(256) void prepare_decode_kernel(
    int32_t* __restrict__ req_to_token, const int64_t* __restrict__ req_pool_indices,
    int64_t* __restrict__ seq_lens, const int64_t* __restrict__ out_cache_loc,
    int64_t* __restrict__ positions, int64_t stride, int64_t bs) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= bs) return;
  const int64_t old_len = seq_lens[b];
  const int64_t new_len = old_len + 1;
  seq_lens[b] = new_len;
  positions[b] = new_len - 1 > 0 ? new_len - 1 : 0;
  req_to_token[req_pool_indices[b] * stride + old_len] = (int32_t)out_cache_loc[b];
}

template <typename PT, typename LT>
static void kv_indices_launch(const int32_t* r2t, const void* pool, const void* lens,
                              const int32_t* indptr, const int32_t* start, int32_t* out,
                              int64_t stride, int64_t bs, hipStream_t st) {
  hipLaunchKernelGGL((create_kv_indices_kernel<PT, LT>), dim3((unsigned)bs), dim3(256), 0, st, r2t,
                     (const PT*)pool, (const LT*)lens, indptr, start, out, stride);
}

// Paged slot allocation (page_size > 1): one workgroup per request.  Request b's new tokens go
// (1) into the rest of its last, partially filled page (slots last_loc + 1 ...), (2) into whole new
// pages and (3) into the head of one more new page, the new pages being taken in request order
// from the front of free_pages.  Both exclusive prefix sums (extend tokens and new pages of the
// requests before b) are computed by the workgroup itself, so the launch has no dependency chain.
// totals[0] = new pages of the whole batch, totals[1] = extend tokens of the whole batch.
__global__ __launch_bounds__(256) void alloc_extend_kernel(
    const int64_t* __restrict__ pre_lens, const int64_t* __restrict__ seq_lens,
    const int64_t* __restrict__ last_loc, const int64_t* __restrict__ free_pages,
    int64_t* __restrict__ out_indices, int64_t* __restrict__ totals, int bs, int64_t page) {
  const int b = blockIdx.x;
  __shared__ int64_t red[2][4];
  int64_t tok = 0, pages = 0;
  for (int i = threadIdx.x; i <= b; i += 256) {
    const int64_t s = seq_lens[i], pr = pre_lens[i];
    tok += s - pr;
    pages += (s + page - 1) / page - (pr + page - 1) / page;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    tok += __shfl_xor(tok, o, 64);
    pages += __shfl_xor(pages, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = tok;
    red[1][threadIdx.x >> 6] = pages;
  }
  __syncthreads();
  const int64_t tok_incl = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  const int64_t pages_incl = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const int64_t seq = seq_lens[b], pre = pre_lens[b];
  const int64_t my_pages = (seq + page - 1) / page - (pre + page - 1) / page;
  const int64_t out0 = tok_incl - (seq - pre), page0 = pages_incl - my_pages;
  if (b == bs - 1 && threadIdx.x == 0) {
    totals[0] = pages_incl;
    totals[1] = tok_incl;
  }
  const int64_t pre_up = (pre + page - 1) / page * page;
  const int64_t n1 = (seq < pre_up ? seq : pre_up) - pre;                      // rest of the old page
  const int64_t n2 = pre + n1 == seq ? 0 : seq / page * page - pre_up;          // whole new pages
  const int64_t n3 = pre + n1 + n2 == seq ? 0 : seq - seq / page * page;        // head of the last new page
  const int64_t ll = last_loc[b];
  for (int64_t i = threadIdx.x; i < n1; i += 256) out_indices[out0 + i] = ll + 1 + i;
  for (int64_t i = threadIdx.x; i < n2; i += 256)
    out_indices[out0 + n1 + i] = free_pages[page0 + i / page] * page + i % page;
  if (n3 > 0) {
    const int64_t start = free_pages[page0 + my_pages - 1] * page;
    for (int64_t i = threadIdx.x; i < n3; i += 256) out_indices[out0 + n1 + n2 + i] = start + i;
  }
}

// Decode: one new token per request -- the next slot of its last page, or slot 0 of a new page when
// seq_len - 1 is a multiple of the page size.  totals[0] = new pages of the batch.
__global__ __launch_bounds__(256) void alloc_decode_kernel(
    const int64_t* __restrict__ seq_lens, const int64_t* __restrict__ last_loc,
    const int64_t* __restrict__ free_pages, int64_t* __restrict__ out_indices,
    int64_t* __restrict__ totals, int bs, int64_t page) {
  __shared__ int64_t red[4];
  // a single workgroup walks the batch in 256-request blocks carrying the page count forward
  int64_t base = 0;
  for (int b0 = 0; b0 < bs; b0 += 256) {
    const int b = b0 + threadIdx.x;
    const int64_t seq = b < bs ? seq_lens[b] : 1;
    const int64_t need = b < bs ? (seq + page - 1) / page - (seq - 1 + page - 1) / page : 0;
    // inclusive scan over the 256 lanes of this block
    int64_t incl = need;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t up = __shfl_up(incl, o, 64);
      if ((threadIdx.x & 63) >= o) incl += up;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = incl;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += red[w];
    const int64_t excl = base + woff + incl - need;
    if (b < bs) out_indices[b] = need ? free_pages[excl] * page : last_loc[b] + 1;
    base += red[0] + red[1] + red[2] + red[3];
  }
  if (threadIdx.x == 0) totals[0] = base;
}

// Five int64 vectors of n elements copied in one launch (the static inputs of a captured decode step).
struct Copy5 {
  int64_t* dst[5];
  const int64_t* src[5];
};
__global__ __launch_bounds__(256) void copy5_kernel(Copy5 c, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) c.dst[blockIdx.y][i] = c.src[blockIdx.y][i];
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_create_kv_indices(const int32_t* req_to_token, const void* req_pool_indices,
                            int pool_itype, const void* page_kernel_lens, int lens_itype,
                            const int32_t* kv_indptr, const int32_t* kv_start_idx,
                            int32_t* kv_indices, int64_t req_to_token_stride, int64_t bs,
                            void* stream) {
  MI355_REQUIRE(req_to_token && req_pool_indices && page_kernel_lens && kv_indptr && kv_indices,
                MI355_ERR_INVALID, "create_kv_indices: null pointer");
  MI355_REQUIRE(bs >= 0 && req_to_token_stride > 0, MI355_ERR_INVALID,
                "create_kv_indices: bad shape");
  MI355_REQUIRE((pool_itype == MI355_I32 || pool_itype == MI355_I64) &&
                    (lens_itype == MI355_I32 || lens_itype == MI355_I64),
                MI355_ERR_UNSUPPORTED, "create_kv_indices: index types must be int32/int64");
  if (bs == 0) return MI355_OK;
  hipStream_t st = (hipStream_t)stream;
  if (pool_itype == MI355_I32 && lens_itype == MI355_I32)
    kv_indices_launch<int32_t, int32_t>(req_to_token, req_pool_indices, page_kernel_lens, kv_indptr,
                                        kv_start_idx, kv_indices, req_to_token_stride, bs, st);
  else if (pool_itype == MI355_I32)
    kv_indices_launch<int32_t, int64_t>(req_to_token, req_pool_indices, page_kernel_lens, kv_indptr,
                                        kv_start_idx, kv_indices, req_to_token_stride, bs, st);
  else if (lens_itype == MI355_I32)
    kv_indices_launch<int64_t, int32_t>(req_to_token, req_pool_indices, page_kernel_lens, kv_indptr,
                                        kv_start_idx, kv_indices, req_to_token_stride, bs, st);
  else
    kv_indices_launch<int64_t, int64_t>(req_to_token, req_pool_indices, page_kernel_lens, kv_indptr,
                                        kv_start_idx, kv_indices, req_to_token_stride, bs, st);
  MI355_CHECK_LAUNCH("create_kv_indices");
  return MI355_OK;
}

int mi355_write_req_to_token(int32_t* req_to_token, const int64_t* req_pool_indices,
                             const int64_t* pre_lens, const int64_t* seq_lens,
                             const int64_t* extend_lens, const int64_t* out_cache_loc,
                             int64_t req_to_token_stride, int64_t bs, void* stream) {
  MI355_REQUIRE(req_to_token && req_pool_indices && pre_lens && seq_lens && extend_lens &&
                    out_cache_loc,
                MI355_ERR_INVALID, "write_req_to_token: null pointer");
  MI355_REQUIRE(bs >= 0 && req_to_token_stride > 0, MI355_ERR_INVALID,
                "write_req_to_token: bad shape");
  if (bs == 0) return MI355_OK;
  hipLaunchKernelGGL(write_req_to_token_kernel, dim3((unsigned)bs), dim3(256), 0,
                     (hipStream_t)stream, req_to_token, req_pool_indices, pre_lens, seq_lens,
                     extend_lens, out_cache_loc, req_to_token_stride);
  MI355_CHECK_LAUNCH("write_req_to_token");
  return MI355_OK;
}

int mi355_compute_position(const int32_t* extend_prefix_lens, const int32_t* extend_seq_lens,
                           int64_t* positions, int32_t* extend_start_loc, int64_t bs,
                           void* stream) {
  MI355_REQUIRE(extend_seq_lens && positions && extend_start_loc, MI355_ERR_INVALID,
                "compute_position: null pointer");
  MI355_REQUIRE(bs >= 0, MI355_ERR_INVALID, "compute_position: bad shape");
  if (bs == 0) return MI355_OK;
  hipLaunchKernelGGL(compute_position_kernel, dim3((unsigned)bs), dim3(256), 0,
                     (hipStream_t)stream, extend_prefix_lens, extend_seq_lens, positions,
                     extend_start_loc);
  MI355_CHECK_LAUNCH("compute_position");
  return MI355_OK;
}

int mi355_prepare_decode(int32_t* req_to_token, const int64_t* req_pool_indices,
                         int64_t* seq_lens, const int64_t* out_cache_loc, int64_t* positions,
                         int64_t req_to_token_stride, int64_t bs, void* stream) {
  MI355_REQUIRE(req_to_token && req_pool_indices && seq_lens && out_cache_loc && positions,
                MI355_ERR_INVALID, "prepare_decode: null pointer");
  MI355_REQUIRE(bs >= 0 && req_to_token_stride > 0, MI355_ERR_INVALID, "prepare_decode: bad shape");
  if (bs == 0) return MI355_OK;
  hipLaunchKernelGGL(prepare_decode_kernel, dim3((unsigned)((bs + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, req_to_token, req_pool_indices, seq_lens, out_cache_loc,
                     positions, req_to_token_stride, bs);
  MI355_CHECK_LAUNCH("prepare_decode");
  return MI355_OK;
}

int mi355_alloc_extend(const int64_t* prefix_lens, const int64_t* seq_lens, const int64_t* last_loc,
                       const int64_t* free_pages, int64_t* out_indices, int64_t* totals, int64_t bs,
                       int64_t page_size, void* stream) {
  // free_pages may be null: an empty free list is legal when no request needs a new page
  MI355_REQUIRE(prefix_lens && seq_lens && last_loc && out_indices && totals, MI355_ERR_INVALID,
                "alloc_extend: null pointer");
  MI355_REQUIRE(bs >= 0 && bs <= 65535 && page_size >= 1, MI355_ERR_INVALID, "alloc_extend: bad shape");
  if (bs == 0) return MI355_OK;
  hipLaunchKernelGGL(alloc_extend_kernel, dim3((unsigned)bs), dim3(256), 0, (hipStream_t)stream, prefix_lens,
                     seq_lens, last_loc, free_pages, out_indices, totals, (int)bs, page_size);
  MI355_CHECK_LAUNCH("alloc_extend");
  return MI355_OK;
}

int mi355_alloc_decode(const int64_t* seq_lens, const int64_t* last_loc, const int64_t* free_pages,
                       int64_t* out_indices, int64_t* totals, int64_t bs, int64_t page_size, void* stream) {
  MI355_REQUIRE(seq_lens && last_loc && out_indices && totals, MI355_ERR_INVALID,
                "alloc_decode: null pointer");
  MI355_REQUIRE(bs >= 0 && bs <= (1 << 24) && page_size >= 1, MI355_ERR_INVALID, "alloc_decode: bad shape");
  if (bs == 0) return MI355_OK;
  hipLaunchKernelGGL(alloc_decode_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, seq_lens, last_loc,
                     free_pages, out_indices, totals, (int)bs, page_size);
  MI355_CHECK_LAUNCH("alloc_decode");
  return MI355_OK;
}

int mi355_copy_decode_inputs(int64_t* const* dst, const int64_t* const* src, int count, int64_t n, void* stream) {
  MI355_REQUIRE(dst && src && count >= 1 && count <= 5 && n >= 0, MI355_ERR_INVALID, "copy_decode_inputs: bad arguments");
  if (n == 0) return MI355_OK;
  Copy5 c;
  for (int i = 0; i < 5; ++i) {
    c.dst[i] = dst[i < count ? i : 0];
    c.src[i] = src[i < count ? i : 0];
    MI355_REQUIRE(c.dst[i] && c.src[i], MI355_ERR_INVALID, "copy_decode_inputs: null pointer");
  }
  hipLaunchKernelGGL(copy5_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)count), dim3(256), 0, (hipStream_t)stream, c, n);
  MI355_CHECK_LAUNCH("copy_decode_inputs");
  return MI355_OK;
}

}  // extern "C"
