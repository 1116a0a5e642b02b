// Which LDS byte does each lane of ds_read_b64_tr_b8 receive?  Lane l supplies address 64 * l.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned char* out, int hi) {
  __shared__ unsigned char lds[4096];
  for (int a = threadIdx.x; a < 4096; a += 64) lds[a] = hi ? (a >> 8) : (a & 0xff);
  __syncthreads();
  i32x2 r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((i32x2 __attribute__((address_space(3)))*)(lds + 64 * threadIdx.x));
  *reinterpret_cast<i32x2*>(out + 8 * threadIdx.x) = r;
}
int main() {
  unsigned char *d, lo[512], hi[512];
  hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 0); hipMemcpy(lo, d, 512, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1); hipMemcpy(hi, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 8; ++j) { int a = hi[8 * l + j] * 256 + lo[8 * l + j]; printf(" (L%d+%d)", a / 64, a % 64); }
    printf("\n");
  }
  return 0;
}
