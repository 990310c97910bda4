// fp64 MFMA rate on varied operand values (clock under load), VGPR-form accumulators, 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NW>
__global__ void __launch_bounds__(256 * NW, 2) k(double* out, const double* in, int iters) {
  d4 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = d4{0, 0, 0, 0};
  double a[4], b[4];
  for (int j = 0; j < 4; ++j) { a[j] = in[threadIdx.x * 4 + j]; b[j] = in[1024 + threadIdx.x * 4 + j]; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j & 3], b[(j + i) & 3], acc[j], 0, 0, 0);
  }
  double s = 0;
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *out, *in; CK(hipMalloc(&out, 1 << 24)); CK(hipMalloc(&in, 1 << 16));
  double h[8192]; unsigned st = 12345;
  for (int z = 0; z < 2; ++z) {
    for (int i = 0; i < 8192; ++i) { st = st * 1664525u + 1013904223u; h[i] = z ? ((double)(st >> 8) / (1 << 24) - 0.5) : 1.0; }
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 40000;
    for (int nw = 1; nw <= 2; ++nw) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (nw == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, in, iters);
        else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, in, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("%s operands, %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s\n", z ? "random" : "constant", nw, ms,
                        256.0 * 4 * nw * iters * 8 * 2048.0 / ms * 1e-9);
      }
    }
  }
  return 0;
}
