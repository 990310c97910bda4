// lbfgsb.hpp — bounded limited-memory BFGS minimiser (host side).
//
// Replaces the reference's call into NLopt's `Algorithm::Lbfgs` with lower/upper bounds and maxeval = 150
// (src/util/gradmin.rs:35-60).  NLopt's Luksan code is a third-party dependency that is not vendored in the
// reference; its iterate sequence is not pinned by any reference test except "a linear objective ends in the
// bound corner" (gradmin.rs:62-101), which this implementation reproduces exactly.  Method: projected L-BFGS —
// active set from the sign of the gradient at the bounds, two-loop recursion on the free variables, projected
// backtracking (Armijo) line search with safeguarded quadratic interpolation.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <vector>

#include "lbfgs_step.hpp"

namespace hbegp {

struct LbfgsOptions {
  int maxeval = 150;
  int memory = 10;
  double pgtol = 1e-7;    // stop when the projected gradient (inf-norm) is below pgtol * max(1, |f|)
  double ftol = 1e-13;    // stop when two successive iterates improve f by less than ftol * max(1, |f|)
  bool fixed_work = false;  // keep evaluating (at the incumbent) until maxeval is used up
};

struct LbfgsResult {
  double f = std::numeric_limits<double>::infinity();
  int nevals = 0;
  int iterations = 0;
  bool converged = false;
};

// objective(x, grad) -> f ; may return +inf (failed evaluation; grad then ignored)
using Objective = std::function<double(const double* x, double* grad)>;

// The method written as nested loops around the objective: the form the state machine of lbfgs_step.hpp was derived from.  Kept
// as the reference of tests/cpp/test_lbfgs_step.cpp (the two must produce the same iterates bit for bit); production code
// calls lbfgsb_minimize() below.
inline LbfgsResult lbfgsb_minimize_loops(const Objective& fun, double* x, const double* lo, const double* hi, int n,
                                         const LbfgsOptions& opt) {
  LbfgsResult res;
  const int m = std::max(1, opt.memory);
  std::vector<double> g(n), xn(n), gn(n), d(n), pg(n), q(n);
  std::vector<std::vector<double>> S, Y;
  std::vector<double> rho;
  auto clip = [&](double* v) {
    for (int i = 0; i < n; ++i) v[i] = std::min(std::max(v[i], lo[i]), hi[i]);
  };
  auto evaluate = [&](const double* xx, double* gg) {
    ++res.nevals;
    double f = fun(xx, gg);
    if (std::isnan(f)) f = std::numeric_limits<double>::infinity();
    return f;
  };
  auto burn = [&](const double* xx) {  // fixed-work mode: spend the remaining evaluations at the incumbent
    if (!opt.fixed_work) return;
    std::vector<double> gt(n);
    while (res.nevals < opt.maxeval) evaluate(xx, gt.data());
  };

  clip(x);
  double f = evaluate(x, g.data());
  res.f = f;
  if (!std::isfinite(f)) {
    // The start point failed.  Nothing to build a model on: report +inf like NLopt would report its failure value.
    burn(x);
    return res;
  }

  while (res.nevals < opt.maxeval) {
    // projected gradient and free set
    double pgnorm = 0;
    for (int i = 0; i < n; ++i) {
      const bool at_lo = x[i] <= lo[i] && g[i] > 0, at_hi = x[i] >= hi[i] && g[i] < 0;
      pg[i] = (at_lo || at_hi) ? 0.0 : g[i];
      pgnorm = std::max(pgnorm, std::fabs(pg[i]));
    }
    if (pgnorm <= opt.pgtol * std::max(1.0, std::fabs(f))) {
      res.converged = true;
      break;
    }
    // two-loop recursion on the free variables
    for (int i = 0; i < n; ++i) q[i] = pg[i];
    const int hcount = (int)S.size();
    std::vector<double> a(hcount);
    for (int h = hcount - 1; h >= 0; --h) {
      double sq = 0;
      for (int i = 0; i < n; ++i) sq += S[h][i] * q[i];
      a[h] = rho[h] * sq;
      for (int i = 0; i < n; ++i) q[i] -= a[h] * Y[h][i];
    }
    double gamma = 1.0;
    if (hcount > 0) {
      double sy = 0, yy = 0;
      for (int i = 0; i < n; ++i) {
        sy += S[hcount - 1][i] * Y[hcount - 1][i];
        yy += Y[hcount - 1][i] * Y[hcount - 1][i];
      }
      if (yy > 0) gamma = sy / yy;
    }
    for (int i = 0; i < n; ++i) q[i] *= gamma;
    for (int h = 0; h < hcount; ++h) {
      double yq = 0;
      for (int i = 0; i < n; ++i) yq += Y[h][i] * q[i];
      const double b = rho[h] * yq;
      for (int i = 0; i < n; ++i) q[i] += S[h][i] * (a[h] - b);
    }
    double dg = 0;
    for (int i = 0; i < n; ++i) {
      d[i] = (pg[i] == 0.0) ? 0.0 : -q[i];
      dg += d[i] * g[i];
    }
    if (!(dg < 0)) {  // not a descent direction: fall back to projected steepest descent
      S.clear(); Y.clear(); rho.clear();
      dg = 0;
      for (int i = 0; i < n; ++i) {
        d[i] = -pg[i];
        dg += d[i] * g[i];
      }
    }
    double step = 1.0;
    if (S.empty()) {
      double dn = 0;
      for (int i = 0; i < n; ++i) dn += d[i] * d[i];
      step = std::min(1.0, 1.0 / std::sqrt(std::max(dn, 1e-300)));
    }
    // projected backtracking line search
    bool accepted = false;
    double fn = f;
    while (res.nevals < opt.maxeval) {
      for (int i = 0; i < n; ++i) xn[i] = x[i] + step * d[i];
      clip(xn.data());
      double gs = 0, moved = 0;
      for (int i = 0; i < n; ++i) {
        gs += g[i] * (xn[i] - x[i]);
        moved = std::max(moved, std::fabs(xn[i] - x[i]));
      }
      if (moved == 0.0) break;  // the step no longer changes x
      fn = evaluate(xn.data(), gn.data());
      if (std::isfinite(fn) && fn <= f + 1e-4 * gs) {
        accepted = true;
        break;
      }
      double next = 0.5 * step;
      if (std::isfinite(fn) && gs < 0) {  // quadratic interpolation through f, f'(0)=gs/step, fn
        const double denom = 2.0 * (fn - f - gs);
        if (denom > 0) next = std::min(std::max(-gs * step / denom, 0.1 * step), 0.5 * step);
      }
      step = next;
      if (step < 1e-20) break;
    }
    if (!accepted) {
      if (!S.empty()) {  // retry once from steepest descent with a clean history
        S.clear(); Y.clear(); rho.clear();
        if (res.nevals < opt.maxeval) continue;
      }
      break;
    }
    ++res.iterations;
    // curvature pair
    std::vector<double> s(n), y(n);
    double sy = 0, ss = 0, yy = 0;
    for (int i = 0; i < n; ++i) {
      s[i] = xn[i] - x[i];
      y[i] = gn[i] - g[i];
      sy += s[i] * y[i];
      ss += s[i] * s[i];
      yy += y[i] * y[i];
    }
    if (sy > 1e-10 * std::sqrt(ss * yy) && sy > 0) {
      if ((int)S.size() == m) {
        S.erase(S.begin()); Y.erase(Y.begin()); rho.erase(rho.begin());
      }
      S.push_back(s); Y.push_back(y); rho.push_back(1.0 / sy);
    }
    const double fold = f;
    for (int i = 0; i < n; ++i) {
      x[i] = xn[i];
      g[i] = gn[i];
    }
    f = fn;
    res.f = f;
    if (fold - f <= opt.ftol * std::max(1.0, std::fabs(f))) {
      // tiny improvement: check once more through the projected-gradient test, otherwise stop
      res.converged = true;
      break;
    }
  }
  burn(x);
  return res;
}

// The same run driven through the resumable state machine (lbfgs_step.hpp) -- the implementation the device runs too.
inline LbfgsResult lbfgsb_minimize(const Objective& fun, double* x, const double* lo, const double* hi, int n, const LbfgsOptions& opt) {
  LbfgsResult res;
  if (n > LBFGS_MAXN) return lbfgsb_minimize_loops(fun, x, lo, hi, n, opt);  // (never the GP: it has at most 66 parameters)
  std::vector<double> g(n);
  LbfgsState st;
  lbfgs_begin(st, x, lo, hi, n, opt.maxeval, opt.memory, opt.pgtol, opt.ftol, opt.fixed_work);
  for (int i = 0; i < n; ++i) x[i] = st.x[i];  // clipped start point (also the result if nothing is ever accepted)
  for (;;) {
    const double f = fun(lbfgs_request(st), g.data());
    if (!lbfgs_advance(st, f, g.data())) break;
  }
  for (int i = 0; i < n; ++i) x[i] = st.x[i];
  res.f = st.f;
  res.nevals = st.nevals;
  res.iterations = st.iterations;
  res.converged = st.converged != 0;
  return res;
}

}  // namespace hbegp
