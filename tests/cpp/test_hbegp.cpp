// C++ smoke of include/hbegp.hpp: the reference's simple fit->predict known-answer test (src/gpr/predict.rs:54-99)
// through the C++ mirror.  Built and run by tests/test_gpu_cpp.py on the GPU box.
#include <cmath>
#include <cstdio>
#include <vector>

#include "hbegp.hpp"

int main() {
  using hbegp::FittedKernel;
  hbegp::Context ctx(1);
  const double xs[4] = {0.0, 0.5, 0.5, 1.0}, ys[4] = {0.0, 0.8, 1.2, 2.0};
  hbegp::KernelBounds b{{0.001, 0.1, 0.1}, {1.0, 4.0, 2.0}};
  std::vector<double> theta0 = {std::log(1.0), std::log(3.0), std::log(1.5)};
  std::vector<double> starts;  // 4 restarts, deterministic log-uniform points
  unsigned st = 938274;
  for (int r = 0; r < 4; ++r)
    for (int i = 0; i < 3; ++i) {
      st = st * 1664525u + 1013904223u;
      const double u = (st >> 8) / 16777216.0;
      starts.push_back(std::log(b.lo[i]) + (std::log(b.hi[i]) - std::log(b.lo[i])) * u);
    }
  auto fk = FittedKernel<double>::new_(ctx, xs, ys, 4, 1, 2.5, theta0, b, starts);
  auto copy = fk;  // Clone shares the device model
  const double q[5] = {0.0, 0.25, 0.5, 0.75, 1.0}, want[5] = {0.0, 0.5, 1.0, 1.5, 2.0};
  double mean[5], var[5];
  copy.predict(q, 5, mean, var);
  int bad = 0;
  for (int i = 0; i < 5; ++i) {
    if (std::fabs(mean[i] - want[i]) > 0.1) ++bad;
    if (std::fabs(var[i] - 0.03) > 0.03) ++bad;
  }
  // not positive definite -> the C++ mirror throws where the reference panics (fit.rs:55)
  const double xd[6] = {0.1, 0.2, 0.1, 0.2, 0.5, 0.5}, yd[3] = {1, 2, 3};
  bool threw = false;
  try {
    FittedKernel<double>::extend(ctx, xd, yd, 3, 2, 2.5, {std::log(1e-300), 0.0, 0.0, 0.0});
  } catch (const hbegp::NotPositiveDefinite&) {
    threw = true;
  }
  // incremental extend through the mirror: 200 rows, prior on the first 150 (one kept 128-block), same alpha as from scratch
  {
    const int n = 200, n0 = 150, d = 2;
    std::vector<double> X(n * d), Y(n);
    unsigned s2 = 12345;
    for (int i = 0; i < n * d; ++i) { s2 = s2 * 1664525u + 1013904223u; X[i] = (s2 >> 8) / 16777216.0; }
    for (int i = 0; i < n; ++i) Y[i] = std::sin(3 * X[2 * i]) + X[2 * i + 1];
    const std::vector<double> th = {std::log(1e-2), 0.0, std::log(0.5), std::log(0.7)};
    auto prior = FittedKernel<double>::extend(ctx, X.data(), Y.data(), n0, d, 2.5, th);
    bool inc = false;
    auto a = prior.extend_with(ctx, X.data(), Y.data(), n, &inc);
    auto f = FittedKernel<double>::extend(ctx, X.data(), Y.data(), n, d, 2.5, th);
    if (!inc) ++bad;
    const auto aa = a.alpha(), fa = f.alpha();
    for (int i = 0; i < n; ++i)
      if (std::fabs(aa[i] - fa[i]) > 1e-9 * (1.0 + std::fabs(fa[i]))) { ++bad; break; }
    if (std::fabs(a.lml() - f.lml()) > 1e-9 * std::fabs(f.lml())) ++bad;
  }
  // ABI growth (hbegp.h: hbegp_fit_options.struct_size): a caller compiled against an OLDER, shorter options struct -- here one
  // that ends behind trace_count, as 0.1.1's did -- must be served without the library reading or writing past it.
  {
    struct OldOptions {
      size_t struct_size;
      int maxeval, fixed_work, lbfgs_memory, trace_cap;
      double *trace_theta, *trace_lml, *trace_grad;
      int* trace_run;
      int* trace_count;
    };
    struct {
      OldOptions o;
      void* guard[2];  // where n_evals / n_not_pd of the current struct would lie: wild pointers the library must not see
    } mem;
    static_assert(sizeof(OldOptions) < sizeof(hbegp_fit_options), "the old layout is a strict prefix");
    mem.o = OldOptions{sizeof(OldOptions), 20, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    mem.guard[0] = reinterpret_cast<void*>(0x10);
    mem.guard[1] = reinterpret_cast<void*>(0x18);
    double tb[3], lb = 0;
    hbegp_model* m = nullptr;
    const int rc = hbegp_fit_f64(ctx.get(), xs, ys, 4, 1, 2.5, theta0.data(), b.lo.data(), b.hi.data(), nullptr, 0,
                                 reinterpret_cast<const hbegp_fit_options*>(&mem.o), tb, &lb, &m);
    if (rc != HBEGP_OK || !m || !std::isfinite(lb)) ++bad;
    if (m) hbegp_model_release(m);
    // and a struct whose size field was never set is refused instead of being trusted
    hbegp_fit_options zero{};
    hbegp_model* m2 = nullptr;
    if (hbegp_fit_f64(ctx.get(), xs, ys, 4, 1, 2.5, theta0.data(), b.lo.data(), b.hi.data(), nullptr, 0, &zero, tb, &lb, &m2) != HBEGP_EINVAL) ++bad;
    if (m2) hbegp_model_release(m2);
  }
  std::printf("lml=%.6f amplitude=%.4f ell=%.4f noise=%.5f bad=%d threw=%d alpha0=%.4f\n", fk.lml(), fk.amplitude(), fk.length_scale()[0],
              fk.noise(), bad, (int)threw, fk.alpha()[0]);
  return (bad == 0 && threw) ? 0 : 1;
}
