// Does hipExtStreamCreateWithCUMask restrict kernels (and graph launches) to a CU subset on this system?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin(double* out, int iters) {
  double x = threadIdx.x;
  for (int i = 0; i < iters; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
  double* out; CK(hipMalloc(&out, 1 << 24));
  hipStream_t full, part;
  CK(hipStreamCreate(&full));
  std::vector<uint32_t> mask(8, 0);  // 256 CUs = 8 words
  mask[0] = 0xffffffffu;             // first 32 CUs
  hipError_t e = hipExtStreamCreateWithCUMask(&part, (uint32_t)mask.size(), mask.data());
  printf("hipExtStreamCreateWithCUMask -> %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 0;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (hipStream_t s : {full, part}) {
    hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, s, out, 20000); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, s, out, 20000);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s stream: %.3f ms\n", s == full ? "full" : "masked(32 CUs)", ms);
  }
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(full, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, full, out, 20000);
  CK(hipStreamEndCapture(full, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (hipStream_t s : {full, part}) {
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph on %s stream: %.3f ms\n", s == full ? "full" : "masked(32 CUs)", ms);
  }
  return 0;
}
