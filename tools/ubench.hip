// Micro-benchmarks that ground the design (fp64 MFMA rate, fp64 VALU rate, co-issue, HBM copy, launch gap).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_mfma64(double* out, int iters, double a0, double b0) {
  d4 acc0 = {0,0,0,0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
  }
  d4 s = acc0 + acc1 + acc2 + acc3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void __launch_bounds__(256) k_mfma32(float* out, int iters, float a0, float b0) {
  f4 acc0 = {0,0,0,0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  float a = a0 + threadIdx.x * 1e-6f, b = b0 - threadIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);
  }
  f4 s = acc0 + acc1 + acc2 + acc3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void __launch_bounds__(256) k_valu64(double* out, int iters, double a0, double b0) {
  double x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-9 + j;
  double a = a0, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __builtin_fma(x[j], a, b);
  }
  double s = 0; for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 512-thread WG: waves 0-3 MFMA, waves 4-7 VALU FMA. mfma iters / valu iters chosen by the host.
__global__ void __launch_bounds__(512) k_mixed(double* out, int it_mfma, int it_valu, double a0, double b0) {
  int wave = threadIdx.x >> 6;
  double r = 0;
  if (wave < 4) {
    d4 acc0 = {0,0,0,0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int i = 0; i < it_mfma; ++i) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    }
    d4 s = acc0 + acc1 + acc2 + acc3; r = s[0] + s[1] + s[2] + s[3];
  } else {
    double x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-9 + j;
    for (int i = 0; i < it_valu; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = __builtin_fma(x[j], a0, b0);
    }
    for (int j = 0; j < 8; ++j) r += x[j];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void k_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void k_write(f4* __restrict__ dst, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  f4 v = {1, 2, 3, 4};
  for (; i < n; i += stride) dst[i] = v;
}
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

template <class F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  double* out; CK(hipMalloc(&out, 1 << 26));
  const int iters = 20000;
  for (int wgs : {256, 512, 1024}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_mfma64, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0, 0.5); }, 3);
    double fl = (double)wgs * 4 * iters * 4 * 2048.0;
    printf("mfma_f64_16x16x4: wgs=%d  %.3f ms  %.2f TFLOP/s\n", wgs, ms, fl / ms * 1e-9);
  }
  for (int wgs : {256, 512}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_mfma32, dim3(wgs), dim3(256), 0, 0, (float*)out, iters, 1.0f, 0.5f); }, 3);
    double fl = (double)wgs * 4 * iters * 4 * 2048.0;
    printf("mfma_f32_16x16x4: wgs=%d  %.3f ms  %.2f TFLOP/s\n", wgs, ms, fl / ms * 1e-9);
  }
  for (int wgs : {256, 512, 1024, 2048}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_valu64, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); }, 3);
    double fl = (double)wgs * 256 * iters * 8 * 2.0;
    printf("v_fma_f64: wgs=%d  %.3f ms  %.2f TFLOP/s\n", wgs, ms, fl / ms * 1e-9);
  }
  {
    // mixed: balance so both halves take similar time standalone (mfma 4 instr x 64cyc = 256 cyc/iter; valu 8 x 4cyc = 32 cyc/iter)
    int itm = 20000, itv = 160000;
    float ms = timeit([&] { hipLaunchKernelGGL(k_mixed, dim3(256), dim3(512), 0, 0, out, itm, itv, 1.0000001, 1e-9); }, 3);
    double fm = 256.0 * 4 * itm * 4 * 2048.0, fv = 256.0 * 256 * itv * 8 * 2.0;
    printf("mixed mfma+valu (1 WG/CU): %.3f ms  mfma %.2f + valu %.2f = %.2f TFLOP/s\n", ms, fm / ms * 1e-9, fv / ms * 1e-9, (fm + fv) / ms * 1e-9);
    float ms1 = timeit([&] { hipLaunchKernelGGL(k_mixed, dim3(256), dim3(512), 0, 0, out, itm, 0, 1.0000001, 1e-9); }, 3);
    float ms2 = timeit([&] { hipLaunchKernelGGL(k_mixed, dim3(256), dim3(512), 0, 0, out, 0, itv, 1.0000001, 1e-9); }, 3);
    printf("   mfma-only %.3f ms, valu-only %.3f ms\n", ms1, ms2);
  }
  {
    size_t bytes = (size_t)2 << 30; f4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    float ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, a, b, bytes / 16); }, 5);
    printf("copy 2GiB: %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0 * bytes / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(k_write, dim3(256 * 8), dim3(256), 0, 0, b, bytes / 16); }, 5);
    printf("write 2GiB: %.3f ms  %.2f TB/s\n", ms, 1.0 * bytes / ms * 1e-9);
    size_t small = (size_t)64 << 20;
    ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, a, b, small / 16); }, 20);
    printf("copy 64MiB (L3-resident): %.4f ms  %.2f TB/s (r+w)\n", ms, 2.0 * small / ms * 1e-9);
    CK(hipFree(a)); CK(hipFree(b));
  }
  {
    float ms = timeit([&] { for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, (int*)nullptr); }, 3);
    printf("1000 empty launches: %.3f ms -> %.2f us/launch\n", ms, ms);
    // graph replay of 200 kernels
    hipStream_t s; CK(hipStreamCreate(&s)); hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, (int*)nullptr);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph of 200 empty kernels: %.3f ms per replay -> %.2f us/kernel\n", ms / 10, ms / 10 / 200 * 1000);
  }
  return 0;
}
