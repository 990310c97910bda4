// Do latency-bound launches of one stream (diagonal-block kernel, tiny GEMMs) make progress while another stream keeps
// the chip full of tile-GEMM workgroups?  Times a chain of 64 dependent launches alone and under a GEMM background.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "engine.hpp"
namespace hbegp { void init_kernels(); }
using namespace hbegp;
int main(int argc, char** argv) {
  const int np = 4096, nb = np / 128;
  const int bgn = argc > 1 ? atoi(argv[1]) / 128 : nb;  // background GEMM size in 128-blocks (launch duration knob)
  std::vector<double> h((size_t)np * np, 0.0);
  for (int i = 0; i < np; ++i) { h[(size_t)i * np + i] = 4.0; if (i) h[(size_t)i * np + i - 1] = 0.5; }
  double *A, *B, *C, *W1, *W2, *ld; int* info;
  const size_t bytes = sizeof(double) * h.size();
  hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&C, bytes); hipMalloc(&W1, bytes); hipMalloc(&W2, bytes); hipMalloc(&ld, np * 8);
  hipMalloc(&info, 4); hipMemset(info, 0, 4);
  hipMemcpy(A, h.data(), bytes, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), bytes, hipMemcpyHostToDevice);
  hipMemcpy(W1, h.data(), bytes, hipMemcpyHostToDevice); hipMemset(W2, 0, bytes);
  init_kernels();
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  const int free_cus = argc > 2 ? atoi(argv[2]) : 0;  // keep this many CUs out of the background stream's CU mask
  if (free_cus > 0) {
    std::vector<uint32_t> mask(8, 0);
    for (int c = 0; c < 256 - free_cus; ++c) mask[c / 32] |= 1u << (c % 32);
    if (hipExtStreamCreateWithCUMask(&sa, 8, mask.data()) != hipSuccess) { printf("CU mask failed\n"); return 1; }
    printf("background stream masked to %d CUs\n", 256 - free_cus);
  } else hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  GemmLaunch big{}; big.nops = 1; big.info = info;
  { GemmOp& op = big.op[0]; op.A = A; op.B = B; op.C = C; op.lda = op.ldb = op.ldc = np; op.mi = bgn; op.nj = bgn; op.k0 = 0; op.k1 = bgn; }
  GemmLaunch tiny{}; tiny.nops = 1; tiny.info = info;
  { GemmOp& op = tiny.op[0]; op.A = W1; op.B = W2; op.C = W2; op.lda = op.ldb = op.ldc = np; op.ci0 = 1; op.mi = 1; op.cj0 = 0; op.nj = 1; op.k0 = 0; op.k1 = 1; }
  for (int kind = 0; kind < 2; ++kind)        // 0: diagonal-block kernel, 1: tiny 32-tile GEMM
    for (int bg = 0; bg < 3; ++bg) {          // 0: alone, 1: under 64-tile GEMMs, 2: under 128-tile GEMMs
      hipDeviceSynchronize();
      const int nbg = (int)(12.0 * nb * nb * nb / ((double)bgn * bgn * bgn));  // ~25 ms of background
      hipEvent_t b0, b1; hipEventCreate(&b0); hipEventCreate(&b1);
      if (bg) { hipEventRecord(b0, sa); for (int r = 0; r < nbg; ++r) launch_gemm<double>(big, bg == 1 ? 64 : 128, sa); hipEventRecord(b1, sa); }
      hipEventRecord(e0, sb);
      const int chain = 64;
      for (int r = 0; r < chain; ++r) {
        if (kind == 0) launch_leaf<double>(W1, W2, np, 0, ld, info, sb);
        else launch_gemm<double>(tiny, 32, sb);
      }
      hipEventRecord(e1, sb); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%s chain of %d, background %s: %.1f us per launch\n", kind == 0 ? "diagonal-block" : "tiny GEMM", chain,
             bg == 0 ? "none" : (bg == 1 ? "64-tile GEMMs" : "128-tile GEMMs"), ms * 1e3 / chain);
      hipDeviceSynchronize();
      if (bg) { float bms; hipEventElapsedTime(&bms, b0, b1); printf("   background: %d launches of n=%d, %.3f ms each, %.1f TFLOP/s\n", nbg, bgn * 128, bms / nbg, 2.0 * nbg * (double)(bgn * 128) * (bgn * 128) * (bgn * 128) / bms * 1e-9); }
    }
  return 0;
}
