// Is v_mfma_f32_16x16x4_f32's accumulation round-to-nearest?  Sum K positive products per output element in one
// accumulator chain and compare with the f64 sum: a signed mean error far from 0 (in units of eps * sum) means truncation.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* C, float* Cf, int K) {
  // one wave: A is 16 x K (row-major), B is K x 16; lane l holds A[l&15][k0 + (l>>4)], B[k0 + (l>>4)][l&15]
  const int lane = threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  float fm[4] = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const float a = A[(lane & 15) * K + k0 + (lane >> 4)];
    const float b = B[(k0 + (lane >> 4)) * 16 + (lane & 15)];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  // reference fmaf chain for element (row = 4*(lane>>4)+r, col = lane&15)
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (lane >> 4) + r, col = lane & 15;
    float s = 0;
    for (int kk = 0; kk < K; ++kk) s = fmaf(A[row * K + kk], B[kk * 16 + col], s);
    fm[r] = s;
    C[row * 16 + col] = acc[r];
    Cf[row * 16 + col] = s;
  }
}
int main() {
  const int K = 2048;
  std::vector<float> A(16 * K), B(K * 16);
  srand(1);
  for (auto& v : A) v = 0.5f + (float)rand() / RAND_MAX;  // positive
  for (auto& v : B) v = 0.5f + (float)rand() / RAND_MAX;
  float *dA, *dB, *dC, *dF;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&dF, 1024);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dF, K);
  std::vector<float> C(256), F(256);
  hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(F.data(), dF, 1024, hipMemcpyDeviceToHost);
  double sm = 0, sf = 0, am = 0, af = 0;
  int same = 0;
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) {
      double ex = 0;
      for (int kk = 0; kk < K; ++kk) ex += (double)A[r * K + kk] * (double)B[kk * 16 + c];
      const double em = (C[r * 16 + c] - ex) / ex, ef = (F[r * 16 + c] - ex) / ex;
      sm += em; sf += ef; am += fabs(em); af += fabs(ef);
      same += C[r * 16 + c] == F[r * 16 + c];
    }
  printf("K=%d positive products: MFMA mean signed rel err %+.3e (mean |err| %.3e); fmaf chain %+.3e (%.3e); bitwise equal %d/256; eps=%.2e\n", K,
         sm / 256, am / 256, sf / 256, af / 256, same, 5.96e-8);
  return 0;
}
