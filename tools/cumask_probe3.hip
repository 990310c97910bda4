// Where does workgroup 0 of successive small dispatches land (XCC), per stream?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned where() {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return ((xcc & 0xf) << 16) | ((hw >> 8) & 0xff);
}
__global__ void probe(unsigned* out) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) out[blockIdx.x] = where();
  if (out[0] == 0xffffffffu) lds[threadIdx.x] = 1;
}
static void seq(const char* name, hipStream_t s, int nwg, int reps, int lds, unsigned* d, hipStream_t other = nullptr) {
  printf("%s, %d wg per dispatch:", name, nwg);
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(512), lds, s, d);
    hipStreamSynchronize(s);
    if (other) { hipLaunchKernelGGL(probe, dim3(3), dim3(64), 0, other, d + 64); hipStreamSynchronize(other); }
    unsigned h[8];
    hipMemcpy(h, d, sizeof(unsigned) * (nwg < 8 ? nwg : 8), hipMemcpyDeviceToHost);
    printf(" [");
    for (int i = 0; i < (nwg < 8 ? nwg : 8); ++i) printf("x%u.cu%u.%u%s", h[i] >> 16, (h[i] >> 5) & 7, h[i] & 15, i + 1 < nwg && i < 7 ? " " : "");
    printf("]");
  }
  printf("\n");
}
int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  unsigned* d; hipMalloc(&d, 4096);
  hipStream_t plain, plain2; hipStreamCreateWithFlags(&plain, hipStreamNonBlocking); hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking);
  std::vector<uint32_t> m(8, 0); m[0] = 0xff;  // CU 0 of every XCC
  hipStream_t masked; hipExtStreamCreateWithCUMask(&masked, 8, m.data());
  std::vector<uint32_t> m2(8, 0); m2[0] = 0xff00;
  hipStream_t masked2; hipExtStreamCreateWithCUMask(&masked2, 8, m2.data());
  seq("plain stream", plain, 1, 12, 159 * 1024, d);
  seq("plain stream", plain, 3, 6, 159 * 1024, d);
  seq("plain stream, other stream dispatching 3 wg between", plain, 1, 12, 159 * 1024, d, plain2);
  seq("masked stream (CU 0 of every XCC)", masked, 1, 12, 159 * 1024, d);
  seq("masked stream 2 (CU 1 of every XCC)", masked2, 1, 12, 159 * 1024, d);
  seq("masked stream, other stream dispatching 3 wg between", masked, 1, 12, 159 * 1024, d, plain2);
  seq("masked stream", masked, 8, 3, 159 * 1024, d);
  seq("masked stream", masked, 1, 4, 159 * 1024, d);
  return 0;
}
