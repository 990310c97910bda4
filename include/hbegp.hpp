// hbegp.hpp — header-only C++ mirror of the reference's src/gpr interface over the C ABI (include/hbegp.h).
//
// Names follow the reference: FittedKernel::new_ / ::extend (src/gpr/fit.rs:18-68; `new` is a C++ keyword),
// FittedKernel::predict (src/gpr/predict.rs:7-52), fields kernel parameters / noise / alpha / k_inv / lml (fit.rs:6-12).
// Errors: usage and runtime errors throw hbegp::Error; the two numerical failures the reference panics on
// (extend on a non-positive-definite matrix fit.rs:55, a fit where every evaluation failed fit.rs:161) throw
// hbegp::NotPositiveDefinite.  Copying a FittedKernel shares the device model (ref-counted, like `Clone` in gpr.rs:53).
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "hbegp.h"

namespace hbegp {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& msg) : std::runtime_error("hbegp error " + std::to_string(c) + ": " + msg), code(c) {}
};
struct NotPositiveDefinite : Error {
  using Error::Error;
};
inline void check(int code) {
  if (code == HBEGP_OK) return;
  if (code == HBEGP_NOT_PD || code == HBEGP_ALL_FAILED) throw NotPositiveDefinite(code, hbegp_last_error());
  throw Error(code, hbegp_last_error());
}

class Context {
 public:
  explicit Context(int n_devices = 1, const int* ids = nullptr) { check(hbegp_ctx_create(n_devices, ids, &h_)); }
  ~Context() { if (h_) hbegp_ctx_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  hbegp_ctx* get() const { return h_; }

 private:
  hbegp_ctx* h_ = nullptr;
};

namespace detail {
template <typename A> struct Abi;
template <> struct Abi<double> {
  static int fit(hbegp_ctx* c, const double* x, const double* y, int n, int d, double nu, const double* t0, const double* lo,
                 const double* hi, const double* st, int r, const hbegp_fit_options* o, double* tb, double* lb, hbegp_model** m) {
    return hbegp_fit_f64(c, x, y, n, d, nu, t0, lo, hi, st, r, o, tb, lb, m);
  }
  static int extend(hbegp_ctx* c, const double* x, const double* y, int n, int d, double nu, const double* t, const double* lo,
                    const double* hi, hbegp_model** m) { return hbegp_extend_f64(c, x, y, n, d, nu, t, lo, hi, m); }
  static int extend_from(hbegp_ctx* c, hbegp_model* p, const double* x, const double* y, int n, hbegp_model** m, int* inc) { return hbegp_extend_from_f64(c, p, x, y, n, m, inc); }
  static int predict(hbegp_model* m, const double* xs, int k, double* mean, double* var, int* w) { return hbegp_predict_f64(m, xs, k, mean, var, w); }
  static int get(hbegp_model* m, double* t, double* a, double* ki) { return hbegp_model_get_f64(m, t, a, ki); }
};
template <> struct Abi<float> {
  static int fit(hbegp_ctx* c, const float* x, const float* y, int n, int d, double nu, const double* t0, const double* lo,
                 const double* hi, const double* st, int r, const hbegp_fit_options* o, double* tb, double* lb, hbegp_model** m) {
    return hbegp_fit_f32(c, x, y, n, d, nu, t0, lo, hi, st, r, o, tb, lb, m);
  }
  static int extend(hbegp_ctx* c, const float* x, const float* y, int n, int d, double nu, const double* t, const double* lo,
                    const double* hi, hbegp_model** m) { return hbegp_extend_f32(c, x, y, n, d, nu, t, lo, hi, m); }
  static int extend_from(hbegp_ctx* c, hbegp_model* p, const float* x, const float* y, int n, hbegp_model** m, int* inc) { return hbegp_extend_from_f32(c, p, x, y, n, m, inc); }
  static int predict(hbegp_model* m, const float* xs, int k, float* mean, float* var, int* w) { return hbegp_predict_f32(m, xs, k, mean, var, w); }
  static int get(hbegp_model* m, double* t, float* a, float* ki) { return hbegp_model_get_f32(m, t, a, ki); }
};
}  // namespace detail

// Kernel = ConstantKernel(amplitude) * Matern(nu, length_scale) + noise (gpr.rs:51); bounds in linear space.
struct KernelBounds {
  std::vector<double> lo, hi;  // order [noise, amplitude, ell_1..ell_d]
};

template <typename A>
class FittedKernel {
 public:
  FittedKernel() = default;
  FittedKernel(const FittedKernel& o) : h_(o.h_), n_(o.n_), d_(o.d_), theta_(o.theta_), lml_(o.lml_) { if (h_) hbegp_model_retain(h_); }
  FittedKernel& operator=(FittedKernel o) { swap(o); return *this; }
  FittedKernel(FittedKernel&& o) noexcept { swap(o); }
  ~FittedKernel() { if (h_) hbegp_model_release(h_); }

  // fit.rs:18-31, 71-176.  theta0: log-space start [ln s2, ln c, ln ell..]; starts: n_restarts * p log-space start points.
  static FittedKernel new_(const Context& ctx, const A* x, const A* y, int n, int d, double nu, const std::vector<double>& theta0,
                           const KernelBounds& b, const std::vector<double>& starts, int maxeval = 150) {
    const int p = d + 2;
    if ((int)theta0.size() != p || (int)b.lo.size() != p || (int)b.hi.size() != p || starts.size() % p != 0)
      throw Error(HBEGP_EINVAL, "theta0 / bounds / starts have the wrong length");
    hbegp_fit_options opt = HBEGP_FIT_OPTIONS_INIT;
    opt.maxeval = maxeval;
    FittedKernel fk;
    fk.n_ = n; fk.d_ = d; fk.theta_.resize(p);
    check(detail::Abi<A>::fit(ctx.get(), x, y, n, d, nu, theta0.data(), b.lo.data(), b.hi.data(), starts.empty() ? nullptr : starts.data(),
                              (int)(starts.size() / p), &opt, fk.theta_.data(), &fk.lml_, &fk.h_));
    return fk;
  }
  // fit.rs:33-68
  static FittedKernel extend(const Context& ctx, const A* x, const A* y, int n, int d, double nu, const std::vector<double>& theta,
                             const KernelBounds* b = nullptr) {
    FittedKernel fk;
    fk.n_ = n; fk.d_ = d; fk.theta_.resize(d + 2);
    check(detail::Abi<A>::extend(ctx.get(), x, y, n, d, nu, theta.data(), b ? b->lo.data() : nullptr, b ? b->hi.data() : nullptr, &fk.h_));
    check(detail::Abi<A>::get(fk.h_, fk.theta_.data(), nullptr, nullptr));
    check(hbegp_model_info(fk.h_, nullptr, nullptr, nullptr, nullptr, &fk.lml_));
    return fk;
  }
  // fit.rs:33-68 at this model's theta on data whose leading rows are this model's training rows (minimize.rs:629-644):
  // reuses the factorisation of the unchanged 128-blocks; *incremental (optional) tells whether it could
  FittedKernel extend_with(const Context& ctx, const A* x, const A* y, int n, bool* incremental = nullptr) const {
    FittedKernel fk;
    fk.n_ = n; fk.d_ = d_; fk.theta_.resize(d_ + 2);
    int inc = 0;
    check(detail::Abi<A>::extend_from(ctx.get(), h_, x, y, n, &fk.h_, &inc));
    if (incremental) *incremental = inc != 0;
    check(detail::Abi<A>::get(fk.h_, fk.theta_.data(), nullptr, nullptr));
    check(hbegp_model_info(fk.h_, nullptr, nullptr, nullptr, nullptr, &fk.lml_));
    return fk;
  }
  // predict.rs:7-52: mean (and variance when var != nullptr); returns the number of variances below -sqrt(1e-5)
  int predict(const A* xs, int m, A* mean, A* var) const {
    int warn = 0;
    check(detail::Abi<A>::predict(h_, xs, m, mean, var, &warn));
    return warn;
  }
  double lml() const { return lml_; }
  double noise() const { return std::exp(theta_[0]); }
  double amplitude() const { return std::exp(theta_[1]); }
  std::vector<double> length_scale() const {
    std::vector<double> e(theta_.begin() + 2, theta_.end());
    for (auto& v : e) v = std::exp(v);
    return e;
  }
  std::vector<A> alpha() const { std::vector<A> a(n_); check(detail::Abi<A>::get(h_, nullptr, a.data(), nullptr)); return a; }
  std::vector<A> k_inv() const { std::vector<A> k((size_t)n_ * n_); check(detail::Abi<A>::get(h_, nullptr, nullptr, k.data())); return k; }

 private:
  void swap(FittedKernel& o) { std::swap(h_, o.h_); std::swap(n_, o.n_); std::swap(d_, o.d_); theta_.swap(o.theta_); std::swap(lml_, o.lml_); }
  hbegp_model* h_ = nullptr;
  int n_ = 0, d_ = 0;
  std::vector<double> theta_;
  double lml_ = 0;
};

}  // namespace hbegp
