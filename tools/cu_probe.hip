// Which (XCC, SE, SH, CU) tuples do workgroups land on, and in what order does the dispatcher walk them?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void probe(unsigned* out) {
  const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, all 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xff);      // cu_id[3:0] sh_id[4] se_id[7:5]
  // keep the workgroup resident for a while so that the grid spreads over the whole chip
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 20000) {}
}
int main() {
  const int n = 2048;
  unsigned* d; hipMalloc(&d, n * 4);
  std::vector<unsigned> h(n);
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 65536, 0, d);   // 64 KB LDS: 2 workgroups per CU
  hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, int> cnt;
  for (unsigned v : h) cnt[v]++;
  printf("%zu distinct (xcc,se,sh,cu) tuples\n", cnt.size());
  std::map<unsigned, std::vector<unsigned>> byx;
  for (auto& kv : cnt) byx[kv.first >> 16].push_back(kv.first & 0xffff);
  for (auto& kv : byx) {
    printf("xcc %u: %zu CUs:", kv.first, kv.second.size());
    for (unsigned c : kv.second) printf(" se%u.sh%u.cu%u", (c >> 5) & 7, (c >> 4) & 1, c & 15);
    printf("\n");
  }
  printf("first 32 workgroups: ");
  for (int i = 0; i < 32; ++i) printf("x%u.se%u.cu%u ", h[i] >> 16, (h[i] >> 5) & 7, h[i] & 15);
  printf("\n");
  return 0;
}
