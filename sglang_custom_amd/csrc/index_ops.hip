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

}  // extern "C"
