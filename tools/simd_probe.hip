#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(unsigned* out) {
  extern __shared__ char smem[];
  unsigned hwid = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hwid;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 8 * 8); 
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  for (int rep = 0; rep < 2; ++rep) {
  hipLaunchKernelGGL(k, dim3(4), dim3(512), 160000, 0, d);
  unsigned h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) { printf("wg %d:", b); for (int w = 0; w < 8; ++w) printf(" w%d: wave_id %u simd %u cu %u |", w, h[b*8+w] & 15, (h[b*8+w] >> 4) & 3, (h[b*8+w] >> 8) & 15); printf("\n"); }
  }
  return 0;
}
