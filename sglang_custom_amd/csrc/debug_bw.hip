// Development aid: read-bandwidth probe used to calibrate what a streaming kernel can reach on
// this chip for a given launch geometry (not part of the hot path).
#include "common.cuh"
#include "../../include/mi355_hotpath_dev.h"

namespace mi355 {

// Each workgroup reads a contiguous slice; every wave keeps U 16-byte loads per lane in flight.
template <int U>
__global__ __launch_bounds__(256) void read_bw_kernel(const uint4* __restrict__ src, int64_t nvec,
                                                      uint32_t* __restrict__ sink) {
  const int64_t per_wg = (nvec + gridDim.x - 1) / gridDim.x;
  const int64_t beg = (int64_t)blockIdx.x * per_wg;
  const int64_t end = beg + per_wg < nvec ? beg + per_wg : nvec;
  uint32_t acc = 0;
  for (int64_t i = beg + threadIdx.x; i < end; i += 256 * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + (int64_t)u * 256;
      v[u] = j < end ? src[j] : uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;  // keeps the loads alive
}

// Which XCD each workgroup runs on (hardware register XCC_ID, low 4 bits).
__global__ void xcc_map_kernel(int* __restrict__ out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15);
}

}  // namespace mi355

extern "C" int mi355_debug_xcc_map(void* out, int grid, void* stream) {
  hipLaunchKernelGGL(mi355::xcc_map_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, (int*)out);
  MI355_CHECK_LAUNCH("debug_xcc_map");
  return MI355_OK;
}

extern "C" int mi355_debug_read_bw(const void* src, int64_t bytes, int grid, int unroll, void* sink,
                                   void* stream) {
  using namespace mi355;
  const int64_t nvec = bytes / 16;
  hipStream_t st = (hipStream_t)stream;
  switch (unroll) {
    case 4: hipLaunchKernelGGL((read_bw_kernel<4>), dim3(grid), dim3(256), 0, st, (const uint4*)src, nvec, (uint32_t*)sink); break;
    case 8: hipLaunchKernelGGL((read_bw_kernel<8>), dim3(grid), dim3(256), 0, st, (const uint4*)src, nvec, (uint32_t*)sink); break;
    default: hipLaunchKernelGGL((read_bw_kernel<16>), dim3(grid), dim3(256), 0, st, (const uint4*)src, nvec, (uint32_t*)sink); break;
  }
  MI355_CHECK_LAUNCH("debug_read_bw");
  return MI355_OK;
}
