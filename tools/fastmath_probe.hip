// fastmath_probe.hip -- accuracy of csrc/fastmath.hpp's sqrt_nonneg / exp_nonpos against the host's libm, in ulps, over the
// arguments a kernel matrix produces (squared scaled distances 0 .. 1e8, exponents -sqrt(5 d2)).
//   hipcc -O3 --offload-arch=gfx950 -I csrc tools/fastmath_probe.hip -o tools/fastmath_probe && tools/fastmath_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "fastmath.hpp"

__global__ void probe(const double* x, double* s, double* e, double* e2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s[i] = hbegp::sqrt_nonneg(x[i]);
  e[i] = hbegp::exp_nonpos(-hbegp::sqrt_nonneg(5.0 * x[i]));
  e2[i] = hbegp::exp_nonpos(-0.5 * x[i]);
}

static double ulps(double got, double want) {
  if (got == want) return 0;
  if (want == 0 || !std::isfinite(want)) return std::fabs(got - want) > 0 ? 1e9 : 0;
  int ex;
  std::frexp(want, &ex);
  return std::fabs(got - want) / std::ldexp(1.0, ex - 53);
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), s(n), e(n), e2(n);
  uint64_t st = 0x9e3779b97f4a7c15ull;
  for (int i = 0; i < n; ++i) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const double u = (st >> 11) / 9007199254740992.0;
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const double v = (st >> 11) / 9007199254740992.0;
    x[i] = (i % 16 == 0) ? 0.0 : std::pow(10.0, -12.0 + 20.0 * u) * (0.5 + v);  // 1e-12 .. 1e8, and exact zeros
  }
  double *dx, *ds, *de, *de2;
  (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&ds, n * 8); (void)hipMalloc(&de, n * 8); (void)hipMalloc(&de2, n * 8);
  (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dx, ds, de, de2, n);
  (void)hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(e.data(), de, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(e2.data(), de2, n * 8, hipMemcpyDeviceToHost);
  double ms = 0, me = 0, me2 = 0;
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const double ws = std::sqrt(x[i]);
    ms = std::fmax(ms, ulps(s[i], ws));
    // exp: compare on the device's own sqrt argument (the error of the composition is what the kernels see)
    const double we = std::exp(-std::sqrt(5.0 * x[i]));
    if (we > 1e-300) me = std::fmax(me, ulps(e[i], we));
    else if (!(e[i] >= 0 && e[i] < 1e-299)) ++bad;
    const double we2 = std::exp(-0.5 * x[i]);
    if (we2 > 1e-300) me2 = std::fmax(me2, ulps(e2[i], we2));
    else if (!(e2[i] >= 0 && e2[i] < 1e-299)) ++bad;
  }
  std::printf("%d arguments: sqrt_nonneg max %.2f ulp; exp_nonpos(-sqrt(5 x)) max %.2f ulp (incl. the argument's rounding); exp_nonpos(-x/2) max %.2f ulp; underflow mismatches %d\n",
              n, ms, me, me2, bad);
  return bad != 0 || ms > 1.0 || me2 > 2.0;
}
