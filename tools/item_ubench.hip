#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int S = 130;
__global__ void __launch_bounds__(512) k(long long* out, double* sink, int mode, int nitems) {
  extern __shared__ double As[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m16 = lane & 15, q4 = lane >> 4;
  for (int e = t; e < 144 * S; e += 512) As[e] = 1.0 / (1 + (e % 97));
  __syncthreads();
  const int P0 = (m16 * S + q4), P1 = (q4 * S + m16);
  // active waves: mode 0: wave 1 only; mode 1: waves 1 and 5 (same SIMD); mode 2: waves 1,2,3 ; mode 3: 1,2,3,5,6,7
  bool active = (mode == 0 && wave == 1) || (mode == 1 && (wave == 1 || wave == 5)) || (mode == 2 && wave >= 1 && wave <= 3) || (mode == 3 && wave != 0 && wave != 4);
  long long t0 = 0, t1 = 0;
  if (active) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nitems; ++it) {
      const int i = 4 + ((it + wave) & 3), j = (it + 2 * wave) & 3, kk = (it + 3 * wave) & 7;
      const double* pa = As + (i * 16) * S + kk * 16 + P0;
      const double* pb = As + (j * 16) * S + kk * 16 + P0;
      double* pc = As + (i * 16) * S + j * 16 + P1;
      double af[4], bf[4], cf[4];
      for (int k4 = 0; k4 < 4; ++k4) { af[k4] = pa[4 * k4]; bf[k4] = pb[4 * k4]; }
      for (int r = 0; r < 4; ++r) cf[r] = pc[4 * r * S];
      d4 a0 = {0,0,0,0}, a1 = {0,0,0,0};
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0], bf[0], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[2], bf[2], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1], bf[1], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[3], bf[3], a1, 0, 0, 0);
      for (int r = 0; r < 4; ++r) pc[4 * r * S] = cf[r] - 1e-9 * (a0[r] + a1[r]);
    }
    t1 = __builtin_readcyclecounter();
  }
  __syncthreads();
  if (lane == 0) out[wave] = t1 - t0;
  if (t == 0) sink[0] = As[5];
}
int main() {
  long long* out; double* sink; hipMalloc(&out, 64); hipMalloc(&sink, 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  for (int mode = 0; mode < 4; ++mode) {
    long long h[8];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 160000, 0, out, sink, mode, 32); hipDeviceSynchronize(); }
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d: cycles per item per wave:", mode);
    for (int w = 0; w < 8; ++w) printf(" %.0f", h[w] / 32.0);
    printf("\n");
  }
  return 0;
}
