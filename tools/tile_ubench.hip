// tile_ubench: the task queue's tile function (dag_gemm_tile) alone -- one 512-thread workgroup per CU, every workgroup runs
// `reps` tiles of a given contraction depth back to back, no queue, no hand-off.  Compares the stage-loop forms (PIPE) in ONE
// process, interleaved rounds (median and best), checks every variant's output bit for bit against PIPE 0.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Icsrc tools/tile_ubench.hip -o tools/tile_ubench
//   ./tools/tile_ubench [depth=2048] [reps=8] [rounds=7] [nwg=256]
#include "../csrc/kernels.hip"

#include <vector>
#include <cstring>
#include <algorithm>

using namespace hbegp;

template <typename T, int TA, int TB, int PIPE>
__global__ void __launch_bounds__(512, 2) tile_bench_kernel(T* W1, T* W2, T* W3, int ld, int depth, int reps, int flags,
                                                            long long* cyc, int map) {
  extern __shared__ __align__(16) char smem_raw[];
  const int per_row = ld / TB;
  const int tile = blockIdx.x;
  int row0 = (tile / per_row) * TA % ld, col0 = (tile % per_row) * TB;
  // TILE_UBENCH_MAP (timing rounds only; workgroup t runs on XCD t % 8).  0: tiles in row-major order -- the 8 tiles of one column
  // of tiles sit on ONE XCD (they share their B panel in its L2), the tiles of a row 4 per XCD.  1: no two workgroups of an XCD
  // share an operand panel (A from W2, B from W3; the eight workgroups t = 8q .. 8q+7 share theirs, one per XCD).  2: all
  // workgroups of an XCD run the SAME tile (everything but the first touch comes from the L2).
  if (map == 1) {
    const int q = (tile / 8) % (ld / TA);
    row0 = q * TA;
    col0 = ((q * 5 + 3) % (ld / TB)) * TB;
    flags = DAGF_ABUF | DAGF_B3;
  } else if (map == 2) {
    row0 = (tile % 8) * TA;
    col0 = ((tile % 8) * 3 % (ld / TB)) * TB;
  }
  auto nop = []() {};
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    // shift the k window per repetition so that the tile's operands are not simply L1/L2-resident from the last round
    const int k0 = ((r * depth) % (ld - depth + 1)) / 128 * 128;
    dag_gemm_tile<T, TA, TB, PIPE>(flags, row0, col0, k0, k0 + depth, W1, W2, W3, (T*)nullptr, ld, smem_raw, nop, nop);
    __syncthreads();
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = (long long)__builtin_readcyclecounter() - t0;
}

static int g_map = 0;
template <typename T, int TA, int TB, int PIPE>
static float run(T* W1, T* W2, T* W3, int ld, int depth, int reps, int flags, int nwg, long long* cyc) {
  static bool once = false;
  auto fn = tile_bench_kernel<T, TA, TB, PIPE>;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, DAG_LDS_BYTES);
    once = true;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(nwg), dim3(512), DAG_LDS_BYTES, 0, W1, W2, W3, ld, depth, reps, flags, cyc, g_map);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms;
}

struct Variant {
  const char* name;
  int ta, tb;
  float (*fn)(double*, double*, double*, int, int, int, int, int, long long*);
};

int main(int argc, char** argv) {
  const int depth = argc > 1 ? atoi(argv[1]) : 2048;
  const int reps = argc > 2 ? atoi(argv[2]) : 8;
  const int rounds = argc > 3 ? atoi(argv[3]) : 7;
  const int nwg = argc > 4 ? atoi(argv[4]) : 256;
  const int ld = 4096;
  const size_t ne = (size_t)ld * ld;
  std::vector<double> h(ne);
  unsigned long long sst = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < ne; ++i) {
    sst ^= sst << 13; sst ^= sst >> 7; sst ^= sst << 17;
    h[i] = (double)(sst >> 11) / 9007199254740992.0 - 0.5;
  }
  double *W1, *W2, *W3, *ref;
  long long* cyc;
  hipMalloc(&W1, ne * 8); hipMalloc(&W2, ne * 8); hipMalloc(&W3, ne * 8); hipMalloc(&ref, ne * 8);
  hipMalloc(&cyc, sizeof(long long) * 4096);
  hipMemcpy(W2, h.data(), ne * 8, hipMemcpyHostToDevice);
  hipMemcpy(W3, h.data(), ne * 8, hipMemcpyHostToDevice);
  const int flags = DAGF_A3 | DAGF_B3;  // C(W1) = A(W3) B(W3)^T, both operands stored [outer][k]
  std::vector<Variant> vs = {
      {"128x128 pipe0", 128, 128, run<double, 128, 128, 0>}, {"128x128 pipe1", 128, 128, run<double, 128, 128, 1>},
      {"128x64  pipe0", 128, 64, run<double, 128, 64, 0>},   {"128x64  pipe1", 128, 64, run<double, 128, 64, 1>},
      {"64x64   pipe0", 64, 64, run<double, 64, 64, 0>},     {"64x64   pipe1", 64, 64, run<double, 64, 64, 1>},
  };
  if (getenv("TILE_UBENCH_ABL")) {  // what each part of the loop costs beside the MFMAs (wrong results: the checks below will say so)
    vs = {
        {"128x128 pipe1", 128, 128, run<double, 128, 128, 1>},
        {"128x128 pipe1 no global loads", 128, 128, run<double, 128, 128, 1 + 16 * 1>},
        {"128x128 pipe1 no loads, no LDS stores", 128, 128, run<double, 128, 128, 1 + 16 * 3>},
        {"128x128 pipe1 no loads, stores, barrier", 128, 128, run<double, 128, 128, 1 + 16 * 7>},
        {"128x128 pipe1 MFMAs only", 128, 128, run<double, 128, 128, 1 + 16 * 15>},
        {"128x128 pipe1 no fragment reads only", 128, 128, run<double, 128, 128, 1 + 16 * 8>},
        {"128x128 pipe1 no barrier only", 128, 128, run<double, 128, 128, 1 + 16 * 4>},
        {"128x64  pipe1", 128, 64, run<double, 128, 64, 1>},
        {"128x64  pipe1 MFMAs only", 128, 64, run<double, 128, 64, 1 + 16 * 15>},
    };
  }
  // correctness: every variant computes the WHOLE ld x ld product once (as many workgroups as tiles) and must reproduce
  // variant 0 bit for bit (the bits depend neither on the stage-loop form nor on the tile shape: one k-ascending chain per element)
  std::vector<double> o0(ne), o1(ne);
  for (size_t v = 0; v < vs.size(); ++v) {
    hipMemset(W1, 0, ne * 8);
    vs[v].fn(W1, W2, W3, ld, depth, 1, flags, (ld / vs[v].ta) * (ld / vs[v].tb), cyc);
    hipMemcpy(v == 0 ? o0.data() : o1.data(), W1, ne * 8, hipMemcpyDeviceToHost);
    if (v == 0) {
      size_t nz = 0;
      for (size_t i = 0; i < ne; ++i) nz += o0[i] != 0.0;
      printf("%s: reference, %zu of %zu outputs nonzero\n", vs[0].name, nz, ne);
    } else {
      printf("%s vs %s: %s\n", vs[v].name, vs[0].name, memcmp(o0.data(), o1.data(), ne * 8) == 0 ? "bitwise equal" : "DIFFERENT");
    }
  }
  // timing: interleaved rounds
  g_map = getenv("TILE_UBENCH_MAP") ? atoi(getenv("TILE_UBENCH_MAP")) : 0;
  printf("timing with TILE_UBENCH_MAP=%d\n", g_map);
  std::vector<std::vector<float>> ms(vs.size());
  std::vector<std::vector<double>> kc(vs.size());
  std::vector<long long> hc(4096);
  for (int r = 0; r < rounds + 1; ++r)
    for (size_t v = 0; v < vs.size(); ++v) {
      const float m = vs[v].fn(W1, W2, W3, ld, depth, reps, flags, nwg, cyc);
      hipMemcpy(hc.data(), cyc, sizeof(long long) * nwg, hipMemcpyDeviceToHost);
      std::sort(hc.begin(), hc.begin() + nwg);
      if (r > 0) { ms[v].push_back(m); kc[v].push_back((double)hc[nwg / 2]); }
    }
  for (size_t v = 0; v < vs.size(); ++v) {
    std::sort(ms[v].begin(), ms[v].end());
    std::sort(kc[v].begin(), kc[v].end());
    const double flop = 2.0 * vs[v].ta * vs[v].tb * (double)depth * reps * nwg;
    const double med = ms[v][ms[v].size() / 2], best = ms[v][0];
    const double cyc_tile = kc[v][kc[v].size() / 2] / reps;
    const double ideal = (double)vs[v].ta * vs[v].tb * depth * 2.0 / 128.0;  // cycles at 128 flop per clock per CU
    printf("%s depth %d: median %.3f ms %.1f TFLOP/s, best %.3f ms %.1f TFLOP/s; in-kernel %.0f cycles per tile = %.1f %% of the MFMA rate (%.2f us at 2.4 GHz)\n",
           vs[v].name, depth, med, flop / med * 1e-9, best, flop / best * 1e-9, cyc_tile, 100.0 * ideal / cyc_tile, cyc_tile / 2400.0);
  }
  return 0;
}
