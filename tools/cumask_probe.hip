// Feasibility probe for the chain/bulk split of the task-queue factorisation:
//  1. which physical CUs does a stream created with hipExtStreamCreateWithCUMask use (single bits, complement masks)?
//  2. is the mask honoured by hipGraphLaunch on that stream?
//  3. while the complement-masked stream keeps the chip full of long-running workgroups, does a full-LDS workgroup on the
//     single-CU stream start at once?
//  4. do two 512-thread workgroups (58 KB LDS, <=128 VGPRs) of DIFFERENT streams share a CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned where() {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return ((xcc & 0xf) << 16) | ((hw >> 8) & 0xff);  // cu_id[3:0] sh_id[4] se_id[7:5]
}
__global__ void probe(unsigned* out, long long ticks) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) out[blockIdx.x] = where();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (ticks < 0) lds[threadIdx.x] = 1;
}
__global__ void stamp(unsigned long long* out) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) { out[0] = wall_clock64(); out[1] = where(); }
  if (out[0] == 1) lds[threadIdx.x] = 1;
}
__global__ void now(unsigned long long* out) { out[0] = wall_clock64(); }

static std::set<unsigned> run_probe(hipStream_t s, int nwg, int threads, int lds, unsigned* d, long long ticks = 20000) {
  hipLaunchKernelGGL(probe, dim3(nwg), dim3(threads), lds, s, d, ticks);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(nwg);
  hipMemcpy(h.data(), d, nwg * 4, hipMemcpyDeviceToHost);
  return std::set<unsigned>(h.begin(), h.end());
}
static void show(const char* what, const std::set<unsigned>& s) {
  printf("%s: %zu CUs", what, s.size());
  if (s.size() <= 8) for (unsigned c : s) printf("  x%u.se%u.sh%u.cu%u", c >> 16, (c >> 5) & 7, (c >> 4) & 1, c & 15);
  printf("\n");
}

int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stamp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  unsigned* d; CK(hipMalloc(&d, 1 << 20));
  unsigned long long* dts; CK(hipMalloc(&dts, 256));
  hipStream_t full; CK(hipStreamCreateWithFlags(&full, hipStreamNonBlocking));
  show("unmasked stream, 4096 wg", run_probe(full, 4096, 256, 65536, d));
  // single bits
  for (int b : {0, 1, 2, 8, 37, 128, 255}) {
    std::vector<uint32_t> m(8, 0);
    m[b / 32] = 1u << (b % 32);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.data());
    if (e != hipSuccess) { printf("bit %d: create failed: %s\n", b, hipGetErrorString(e)); continue; }
    char nm[64]; snprintf(nm, sizeof nm, "mask bit %d, 64 wg x 512 thr, 159 KB", b);
    show(nm, run_probe(s, 64, 512, 159 * 1024, d, 2000));
    hipStreamDestroy(s);
  }
  // three chain CUs (bits 0,1,2) and the complement
  std::vector<uint32_t> mc(8, 0xffffffffu), m0(8, 0), m1(8, 0), m2(8, 0);
  mc[0] = ~7u; m0[0] = 1; m1[0] = 2; m2[0] = 4;
  hipStream_t sc, s0, s1, s2, sc2;
  CK(hipExtStreamCreateWithCUMask(&sc, 8, mc.data()));
  CK(hipExtStreamCreateWithCUMask(&sc2, 8, mc.data()));
  CK(hipExtStreamCreateWithCUMask(&s0, 8, m0.data()));
  CK(hipExtStreamCreateWithCUMask(&s1, 8, m1.data()));
  CK(hipExtStreamCreateWithCUMask(&s2, 8, m2.data()));
  std::set<unsigned> comp = run_probe(sc, 4096, 256, 65536, d);
  show("complement of bits 0-2, 4096 wg", comp);
  std::set<unsigned> c0 = run_probe(s0, 8, 512, 159 * 1024, d, 2000), c1 = run_probe(s1, 8, 512, 159 * 1024, d, 2000), c2 = run_probe(s2, 8, 512, 159 * 1024, d, 2000);
  show("bit 0", c0); show("bit 1", c1); show("bit 2", c2);
  int overlap = 0;
  for (auto& cs : {c0, c1, c2}) for (unsigned c : cs) overlap += comp.count(c);
  printf("chain CUs found inside the complement set: %d (want 0)\n", overlap);
  // graph launch on the masked stream
  {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(full, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 65536, full, d, 20000LL);
    CK(hipStreamEndCapture(full, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, sc)); CK(hipStreamSynchronize(sc));
    std::vector<unsigned> h(4096);
    hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> gs(h.begin(), h.end());
    show("graph (captured on the unmasked stream) launched on the complement stream", gs);
    CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
    hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    show("same graph launched on the bit-0 stream", std::set<unsigned>(h.begin(), h.end()));
  }
  // 3. chip full of long-running workgroups on the complement stream (2 x 512 threads x 64 KB per CU, 20 ms each);
  //    how long until a full-LDS workgroup on the bit-0 stream runs?
  {
    unsigned* d2; CK(hipMalloc(&d2, 1 << 20));
    hipLaunchKernelGGL(probe, dim3(506), dim3(512), 60 * 1024, sc, d2, 2000000LL);  // 20 ms at 100 MHz
    hipLaunchKernelGGL(now, dim3(1), dim3(64), 0, s1, dts + 8);
    CK(hipStreamSynchronize(s1));
    hipLaunchKernelGGL(stamp, dim3(1), dim3(512), 159 * 1024, s0, dts);
    CK(hipStreamSynchronize(s0));
    unsigned long long h[16];
    CK(hipMemcpy(h, dts, 128, hipMemcpyDeviceToHost));
    printf("full-LDS workgroup on the bit-0 stream started %.1f us after a reference kernel on bit-1 (chip full on the complement stream): cu x%llu.se%llu.cu%llu\n",
           (double)(h[0] - h[8]) / 100.0, h[1] >> 16, (h[1] >> 5) & 7, h[1] & 15);
    CK(hipStreamSynchronize(sc));
    std::vector<unsigned> hh(506);
    hipMemcpy(hh.data(), d2, 506 * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> cnt;
    for (unsigned v : hh) cnt[v]++;
    int two = 0, one = 0, more = 0;
    for (auto& kv : cnt) (kv.second == 2 ? two : (kv.second == 1 ? one : more))++;
    printf("506 x (512 thr, 60 KB) on the complement stream: %zu CUs used, %d hold 2, %d hold 1, %d hold more\n", cnt.size(), two, one, more);
  }
  // 4. two streams, 253 workgroups each (512 thr, 60 KB): do they share CUs?
  {
    unsigned *da, *db; CK(hipMalloc(&da, 4096)); CK(hipMalloc(&db, 4096));
    hipLaunchKernelGGL(probe, dim3(253), dim3(512), 60 * 1024, sc, da, 200000LL);
    hipLaunchKernelGGL(probe, dim3(253), dim3(512), 60 * 1024, sc2, db, 200000LL);
    CK(hipStreamSynchronize(sc)); CK(hipStreamSynchronize(sc2));
    std::vector<unsigned> ha(253), hb(253);
    hipMemcpy(ha.data(), da, 253 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), db, 253 * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> sa(ha.begin(), ha.end()), sb(hb.begin(), hb.end());
    int shared = 0;
    for (unsigned c : sa) shared += sb.count(c);
    printf("two streams x 253 wg: stream A on %zu CUs, stream B on %zu CUs, %d CUs host both\n", sa.size(), sb.size(), shared);
  }
  return 0;
}
