// Stage times of the single-launch evaluation (small_eval_kernel) from its cycle stamps.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icsrc -c tools/small_eval_bench.cpp -o build/small_eval_bench.o && hipcc --offload-arch=gfx950 build/small_eval_bench.o build/kernels.o -o tools/small_eval_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "engine.hpp"
namespace hbegp { void init_kernels(); void read_leaf_stamps(long long* out); }
using namespace hbegp;
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 128, d = argc > 2 ? atoi(argv[2]) : 8;
  std::vector<double> X((size_t)n * d), y(128, 0.0);
  for (int i = 0; i < n; ++i) { for (int k = 0; k < d; ++k) X[(size_t)i * d + k] = fmod(0.37 * i + 0.11 * k * (i + 1), 1.0); y[i] = sin(3.0 * X[(size_t)i * d]); }
  EvalParams P{}; P.noise = 0.1; P.amp = 1.0; for (int k = 0; k < d; ++k) P.ell[k] = 0.7;
  double *dX, *dy, *W2, *ld, *Kinv, *alpha; EvalParams* dP; EvalOut* dOut;
  hipMalloc(&dX, X.size() * 8); hipMalloc(&dy, 128 * 8); hipMalloc(&W2, 128 * 128 * 8); hipMalloc(&ld, 128 * 8); hipMalloc(&Kinv, 128 * 128 * 8);
  hipMalloc(&alpha, 128 * 8); hipMalloc(&dP, sizeof(P)); hipMalloc(&dOut, sizeof(EvalOut));
  hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), 128 * 8, hipMemcpyHostToDevice);
  hipMemcpy(dP, &P, sizeof(P), hipMemcpyHostToDevice); hipMemset(W2, 0, 128 * 128 * 8); hipMemset(dOut, 0, sizeof(EvalOut));
  init_kernels();
  SmallEval g{}; g.X = dX; g.y = dy; g.n = n; g.d = d; g.P = dP; g.W2 = W2; g.ldiag = ld; g.Kinv = Kinv; g.alpha = alpha; g.out = dOut; g.hout = dOut; g.mode = 7;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_small_eval<double>(g, 5, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) launch_small_eval<double>(g, 5, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long st[256]; read_leaf_stamps(st);
  EvalOut out; hipMemcpy(&out, dOut, sizeof(out), hipMemcpyDeviceToHost);
  printf("n=%d d=%d: %.1f us per launch back to back; cycles: kmat %lld | factor+inverse %lld | alpha+lml %lld | K^-1 %lld | gradient %lld | total %lld; lml %.6f info %d done %d\n",
         n, d, ms * 1000 / 50, st[201] - st[200], st[202] - st[201], st[203] - st[202], st[204] - st[203], st[205] - st[204], st[205] - st[200], out.lml, out.info, out.done);
  return 0;
}
