// Times the diagonal-block (leaf) kernel with parts switched off (diagnostic build flags), to see where its time goes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icsrc tools/leaf_bench.cpp build/kernels.o -o tools/leaf_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "engine.hpp"
namespace hbegp { void init_kernels(); void read_leaf_stamps(long long* out); }
using namespace hbegp;
int main() {
  const int nb = 32, np = nb * 128;
  std::vector<double> h((size_t)np * np, 0.0);
  for (int b = 0; b < nb; ++b)
    for (int i = 0; i < 128; ++i)
      for (int j = 0; j < 128; ++j) {
        size_t r = b * 128 + i, c = b * 128 + j;
        h[r * np + c] = (i == j ? 4.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
      }
  double *W1, *W2, *ld; int* info;
  hipMalloc(&W1, sizeof(double) * h.size()); hipMalloc(&W2, sizeof(double) * h.size()); hipMalloc(&ld, sizeof(double) * np);
  hipMalloc(&info, 4); hipMemset(info, 0, 4); hipMemset(W2, 0, sizeof(double) * h.size());
  init_kernels();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int dbg : {0, 1, 2, 4, 3, 5, 6, 7}) {
    hipMemcpy(W1, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    launch_leaf<double>(W1, W2, np, 0, ld, info, 0, dbg);
    hipMemcpy(W1, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int b = 0; b < nb; ++b) launch_leaf<double>(W1, W2, np, b, ld, info, 0, dbg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int hinfo; hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost); hipMemset(info, 0, 4);
    printf("dbg=%d (skip diag=%d phase2=%d mfma-phase1=%d): %.2f us per leaf (info=%d)\n", dbg, dbg & 1, (dbg >> 1) & 1, (dbg >> 2) & 1,
           ms * 1000 / nb, hinfo);
  }
  // per-phase cycle stamps of one block (dbg bit 3)
  for (int extra : {0, 3, 1, 2}) {
  hipMemcpy(W1, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  launch_leaf<double>(W1, W2, np, 0, ld, info, 0, 8 | extra);
  printf("dbg=%d ", extra);
  hipDeviceSynchronize();
  long long st[256];
  read_leaf_stamps(st);
  printf("cycles: load->0");
  for (int p = 0; p < 8; ++p) {
    printf(" | p%d: phase1 %lld", p, st[1 + 2 * p] - (p == 0 ? st[0] : st[2 * p]));
    if (p < 7) printf(" phase2 %lld", st[2 + 2 * p] - st[1 + 2 * p]);
  }
  printf(" | tail %lld | total %lld\n", st[20] - st[15], st[20] - st[0]);
  printf("      MFMA waves in the pool [start after phase start + duration], waves 1 2 3 5 6 7:");
  for (int p = 1; p < 8; ++p) {
    printf(" p%d:", p);
    for (int w : {1, 2, 3, 5, 6, 7}) {
      const long long a = st[64 + (p * 8 + w) * 2], b = st[65 + (p * 8 + w) * 2];
      if (a > 0 && b > 0) printf(" %lld+%lld", a - st[2 * p], b - a); else printf(" -");
    }
  }
  printf("\n");
  printf("      wave 0 per panel [phase start -> loads done | elimination | stores]:");
  for (int p = 0; p < 8; ++p) printf(" p%d: %lld|%lld|%lld", p, st[40 + 3 * p] - (p == 0 ? st[0] : st[2 * p]), st[41 + 3 * p] - st[40 + 3 * p], st[42 + 3 * p] - st[41 + 3 * p]);
  printf("\n");
  }
  return 0;
}
