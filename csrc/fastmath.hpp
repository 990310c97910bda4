// fastmath.hpp -- fp64 sqrt / exp for arguments of known sign (device code; included by kernels.hip and tools/fastmath_probe.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace hbegp {

// ---- fp64 sqrt / exp for the kernel-matrix and gradient passes ---------------------------------------------------------
// kmat / gradtrace / kstar evaluate one sqrt and one exp per matrix entry.  With the library routines (special cases, a true
// division in the polynomial) they were issue-bound on the fp64 pipe: 37 us for the 8.4 M entries of K at n = 4096 where the
// stores alone need 13 us (SQ counters, profiles/r02_pmc_sq.json).  The arguments here are tame -- squared distances >= 0,
// exponents <= 0 -- so both reduce to a short branch-free sequence:
//   sqrt(x), x >= 0:  y = rsq(x) (5e-8) -> one third-order step (1.4e-16, tools/rsq_probe.hip) -> s = x y and one Newton
//                     correction of s; x = 0 -> 0.                                                    (11 instructions)
//   exp(x), x <= 0:   t = rint(x log2 e), r = x - t ln2 (two-part), Taylor to r^13 (|r| <= 0.347: truncation 4e-18), ldexp;
//                     x clamped at -750 (the result is 0 from -745.2 on).                              (19 instructions)
// Measured against exp / sqrt of the host's libm over the arguments a kernel matrix produces: <= 1 ulp (tools/fastmath_probe.hip).
// Contraction is off inside: every caller gets the same bits wherever these are inlined.
__device__ __forceinline__ double sqrt_nonneg(double x) {
#pragma clang fp contract(off)
  double y = __builtin_amdgcn_rsq(x);
  const double g = x * y;
  const double e = __builtin_fma(-g, y, 1.0);
  const double pp = __builtin_fma(e, 0.375, 0.5);
  y = __builtin_fma(y, e * pp, y);
  double sq = x * y;
  const double res = __builtin_fma(-sq, sq, x);
  sq = __builtin_fma(res * 0.5, y, sq);
  return x == 0.0 ? 0.0 : sq;  // rsq(NaN or x < 0) is NaN and stays NaN (a NaN feature must end in NOT_PD, lml.rs:47-50)
}
__device__ __forceinline__ float sqrt_nonneg(float x) { return sqrtf(x); }

__device__ __forceinline__ double exp_nonpos(double x) {
#pragma clang fp contract(off)
  x = x < -750.0 ? -750.0 : x;  // not fmax: a NaN argument stays NaN
  const double t = __builtin_rint(x * 1.4426950408889634074);
  double r = __builtin_fma(-t, 6.93147180369123816490e-01, x);
  r = __builtin_fma(-t, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821614599e-10;             // 1/13!
  p = __builtin_fma(p, r, 2.0876756987868098979e-09);  // 1/12!
  p = __builtin_fma(p, r, 2.5052108385441718775e-08);  // 1/11!
  p = __builtin_fma(p, r, 2.7557319223985890653e-07);  // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985890653e-06);  // 1/9!
  p = __builtin_fma(p, r, 2.4801587301587301587e-05);  // 1/8!
  p = __builtin_fma(p, r, 1.9841269841269841270e-04);  // 1/7!
  p = __builtin_fma(p, r, 1.3888888888888888889e-03);  // 1/6!
  p = __builtin_fma(p, r, 8.3333333333333333333e-03);  // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666666667e-02);  // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666667e-01);  // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)t);
}
__device__ __forceinline__ float exp_nonpos(float x) { return expf(x); }

}  // namespace hbegp
