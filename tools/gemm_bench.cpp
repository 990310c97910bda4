// Intrinsic throughput of the tile GEMM kernel on plain square problems (no triangular structure).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "engine.hpp"
namespace hbegp { void init_kernels(); }
using namespace hbegp;
int main(int argc, char** argv) {
  const int np = argc > 1 ? atoi(argv[1]) : 4096;
  const int nb = np / 128;
  std::vector<double> h((size_t)np * np);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  double *A, *B, *C; int* info;
  hipMalloc(&A, sizeof(double) * h.size()); hipMalloc(&B, sizeof(double) * h.size()); hipMalloc(&C, sizeof(double) * h.size());
  hipMalloc(&info, 4); hipMemset(info, 0, 4);
  hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  init_kernels();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int tile : {128, 64, 32}) {
      GemmLaunch g{}; g.nops = 1; g.info = info;
      GemmOp& op = g.op[0];
      op.A = A; op.B = B; op.C = C; op.lda = op.ldb = op.ldc = np;
      op.a_kmajor = mode == 2; op.b_kmajor = mode >= 1;
      op.mi = nb; op.nj = nb; op.k0 = 0; op.k1 = nb;
      launch_gemm<double>(g, tile, 0); hipDeviceSynchronize();
      hipEventRecord(e0);
      const int reps = 5;
      for (int r = 0; r < reps; ++r) launch_gemm<double>(g, tile, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("n=%d mode=%s tile=%d: %.3f ms  %.1f TFLOP/s\n", np, mode == 0 ? "NT" : (mode == 1 ? "NN" : "TN"), tile, ms,
             2.0 * np * np * (double)np / ms * 1e-9);
    }
  return 0;
}
