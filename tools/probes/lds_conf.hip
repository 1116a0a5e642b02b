// LDS bank-conflict probe for the fp8 MLA ring's three read patterns (run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int m8_row(int r) { return r * 592 + (r >= 16 ? 128 : 0); }
template <int MODE> __global__ void k(int* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32 * 592 + 128 + 8192];
  for (int a = threadIdx.x; a < (int)sizeof(lds) / 4; a += 256) reinterpret_cast<int*>(lds)[a] = a;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, c4 = lane >> 4;
  int acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {   // score reads
      const int k_rd = m8_row(16 * (wave >> 1) + r16) + 288 * (wave & 1) + 8 * c4;
#pragma unroll
      for (int s = 0; s < 9; ++s) { const volatile uint2* vp = reinterpret_cast<const volatile uint2*>(lds + k_rd + 32 * s); uint2 v; v.x = vp->x; v.y = vp->y; acc ^= v.x ^ v.y; }
    } else if (MODE == 1) {   // transposed value reads
      const int vq = r16 >> 1;
      const int v_rd = m8_row(4 * c4 + (vq & 3) + 16 * (vq >> 2)) + 128 * wave + 8 * (r16 & 1);
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((i32x2 __attribute__((address_space(3)))*)(lds + v_rd + 16 * dt));
        acc ^= v[0] ^ v[1];
      }
    } else {   // partial-score exchange
      f32x4* xw = reinterpret_cast<f32x4*>(lds + 32 * 592 + 128 + (it & 1) * 4096);
      f32x4 p = {1.f * acc, 2.f, 3.f, 4.f};
      xw[wave * 64 + lane] = p;
      __syncthreads();
      f32x4 s = xw[lane] + xw[64 + lane] + xw[128 + lane] + xw[192 + lane];
      acc ^= (int)s[0] ^ (int)s[3];
    }
    asm volatile("" : "+v"(acc));
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  int* d; (void)hipMalloc(&d, 256 * 256 * 4);
  hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, 1000);
  hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, 1000);
  hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, 1000);
  (void)hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
