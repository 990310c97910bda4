// Which CU sets do various hipExtStreamCreateWithCUMask masks give on MI355X (SPX mode, 8 XCCs x 32 CUs)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__device__ __forceinline__ unsigned where() {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return ((xcc & 0xf) << 16) | ((hw >> 8) & 0xff);
}
__global__ void probe(unsigned* out, long long ticks) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) out[blockIdx.x] = where();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (ticks < 0) lds[threadIdx.x] = 1;
}
static void test(const char* name, std::vector<int> bits, bool complement) {
  std::vector<uint32_t> m(8, complement ? 0xffffffffu : 0u);
  for (int b : bits) { if (complement) m[b / 32] &= ~(1u << (b % 32)); else m[b / 32] |= 1u << (b % 32); }
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.data());
  if (e != hipSuccess) { printf("%s: create failed %s\n", name, hipGetErrorString(e)); return; }
  std::vector<uint32_t> got(8, 0);
  hipExtStreamGetCUMask(s, 8, got.data());
  unsigned* d; hipMalloc(&d, 1 << 16);
  const int nwg = 2048;
  hipLaunchKernelGGL(probe, dim3(nwg), dim3(512), 159 * 1024, s, d, 300LL);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(nwg);
  hipMemcpy(h.data(), d, nwg * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, int> cnt;
  for (unsigned v : h) cnt[v]++;
  printf("%s%s: %zu CUs; getmask %08x %08x ..;", complement ? "complement of " : "", name, cnt.size(), got[0], got[1]);
  if (cnt.size() <= 24) for (auto& kv : cnt) printf(" x%u.se%u.cu%u", kv.first >> 16, (kv.first >> 5) & 7, kv.first & 15);
  else {
    std::map<unsigned, int> perx;
    for (auto& kv : cnt) perx[kv.first >> 16]++;
    for (auto& kv : perx) printf(" x%u:%d", kv.first, kv.second);
  }
  printf("\n");
  hipFree(d);
  hipStreamDestroy(s);
}
int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  test("bit 0", {0}, false);
  test("bits 0-1", {0, 1}, false);
  test("bits 0-7", {0, 1, 2, 3, 4, 5, 6, 7}, false);
  test("bits 0-15", {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, false);
  test("bits 8-15", {8, 9, 10, 11, 12, 13, 14, 15}, false);
  test("bits 0,8,..,56", {0, 8, 16, 24, 32, 40, 48, 56}, false);
  test("bits 0,32,..,224", {0, 32, 64, 96, 128, 160, 192, 224}, false);
  std::vector<int> w0; for (int i = 0; i < 32; ++i) w0.push_back(i);
  test("bits 0-31", w0, false);
  test("bits 0-7", {0, 1, 2, 3, 4, 5, 6, 7}, true);
  test("bits 0-2", {0, 1, 2}, true);
  test("bits 0-31", w0, true);
  return 0;
}
