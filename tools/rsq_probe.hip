// Relative error of v_rsq_f64 raw, after one and after two Newton steps (the diagonal-block kernel's pivot scaling).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* r3, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  double y = __builtin_amdgcn_rsq(d);
  r0[i] = y;
  y = y + y * (0.5 - 0.5 * d * y * y);   // one Newton step written naively
  r1[i] = y;
  { double e = __builtin_fma(-d * y, y, 1.0); y = __builtin_fma(0.5 * y, e, y); }
  r2[i] = y;
  {  // one third-order (Halley) step from the raw estimate: r (1 + e/2 + 3e^2/8), e = 1 - d r^2
    double r = __builtin_amdgcn_rsq(d);
    const double g = d * r, e = __builtin_fma(-g, r, 1.0), pp = __builtin_fma(e, 0.375, 0.5), q = e * pp;
    r3[i] = __builtin_fma(r, q, r);
  }
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
  double *x, *a, *b, *c, *dd; hipMalloc(&x, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8); hipMalloc(&dd, n * 8);
  hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, a, b, c, dd, n);
  std::vector<double> ra(n), rb(n), rc(n), rd(n); hipMemcpy(rd.data(), dd, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(ra.data(), a, n * 8, hipMemcpyDeviceToHost); hipMemcpy(rb.data(), b, n * 8, hipMemcpyDeviceToHost); hipMemcpy(rc.data(), c, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  for (int i = 0; i < n; ++i) {
    const long double ex = 1.0L / sqrtl((long double)h[i]);
    e0 = fmax(e0, (double)fabsl((ra[i] - ex) / ex)); e1 = fmax(e1, (double)fabsl((rb[i] - ex) / ex)); e2 = fmax(e2, (double)fabsl((rc[i] - ex) / ex)); e3 = fmax(e3, (double)fabsl((rd[i] - ex) / ex));
  }
  printf("max relative error: raw v_rsq_f64 %.3e, +1 Newton %.3e, +2 Newton %.3e, one Halley step %.3e (eps = %.3e)\n", e0, e1, e2, e3, 2.22e-16);
  return 0;
}
