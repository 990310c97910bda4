// The resumable L-BFGS (csrc/lbfgs_step.hpp, the form a persistent kernel can run) must evaluate exactly the points the
// loop form (lbfgsb_minimize_loops) evaluates, in the same order, bit for bit, and end in the same state.
// Build + run (CPU): g++ -O2 -std=c++17 -ffp-contract=off -Icsrc tests/cpp/test_lbfgs_step.cpp -o build/test_lbfgs_step && build/test_lbfgs_step
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "lbfgsb.hpp"

using namespace hbegp;

struct Case {
  const char* name;
  int n;
  std::vector<double> x0, lo, hi;
  LbfgsOptions opt;
  int kind;
};

static double objective(int kind, int n, const double* x, double* g) {
  double f = 0;
  switch (kind) {
    case 0:  // slanted plane (gradmin.rs:75-101): ends in the corner of the box
      for (int i = 0; i < n; ++i) { f += (i + 1) * x[i]; g[i] = i + 1; }
      return f;
    case 1:  // Rosenbrock
      for (int i = 0; i < n; ++i) g[i] = 0;
      for (int i = 0; i + 1 < n; ++i) {
        const double a = x[i + 1] - x[i] * x[i], b = 1 - x[i];
        f += 100 * a * a + b * b;
        g[i] += -400 * x[i] * a - 2 * b;
        g[i + 1] += 200 * a;
      }
      return f;
    case 2:  // convex quadratic with an ill-conditioned Hessian
      for (int i = 0; i < n; ++i) { const double w = std::pow(10.0, 3.0 * i / (n - 1.0)); f += 0.5 * w * (x[i] - 0.3) * (x[i] - 0.3); g[i] = w * (x[i] - 0.3); }
      return f;
    case 3: {  // fails (+inf) outside a ball: the line search has to back off; NaN further out
      double r2 = 0;
      for (int i = 0; i < n; ++i) r2 += x[i] * x[i];
      if (r2 > 9.0) return NAN;
      if (r2 > 4.0) return INFINITY;
      for (int i = 0; i < n; ++i) { f += std::cos(x[i]) + 0.1 * x[i]; g[i] = -std::sin(x[i]) + 0.1; }
      return f;
    }
    default:  // fails at the start point
      for (int i = 0; i < n; ++i) g[i] = 0;
      return INFINITY;
  }
}

int main() {
  std::vector<Case> cases;
  auto add = [&](const char* name, int n, double x0, double lo, double hi, int kind, int maxeval, int memory, bool fixed) {
    Case c{name, n, std::vector<double>(n), std::vector<double>(n, lo), std::vector<double>(n, hi), LbfgsOptions(), kind};
    for (int i = 0; i < n; ++i) c.x0[i] = x0 + 0.37 * std::sin(1.0 + 3.0 * i);
    c.opt.maxeval = maxeval; c.opt.memory = memory; c.opt.fixed_work = fixed;
    cases.push_back(c);
  };
  add("plane", 3, 0.5, -1, 2, 0, 150, 10, false);
  add("plane fixed work", 3, 0.5, -1, 2, 0, 40, 10, true);
  add("rosenbrock", 8, -0.5, -2.5, 2.5, 1, 150, 10, false);
  add("rosenbrock memory 3", 8, -0.5, -2.5, 2.5, 1, 150, 3, false);
  add("rosenbrock 66", 66, -0.5, -2.5, 2.5, 1, 150, 10, true);
  add("rosenbrock tight box", 5, 0.0, -0.25, 0.75, 1, 150, 10, false);
  add("quadratic", 10, 2.0, -5, 5, 2, 150, 10, false);
  add("quadratic maxeval 7", 10, 2.0, -5, 5, 2, 7, 10, true);
  add("quadratic maxeval 1", 10, 2.0, -5, 5, 2, 1, 10, false);
  add("failing region", 4, 0.4, -5, 5, 3, 150, 10, false);
  add("failing region fixed", 4, 0.4, -5, 5, 3, 60, 10, true);
  add("failing start", 4, 0.4, -5, 5, 4, 20, 10, true);
  add("failing start, no burn", 4, 0.4, -5, 5, 4, 20, 10, false);
  add("start outside the box", 6, 9.0, -2.5, 2.5, 1, 150, 10, false);
  int bad = 0;
  for (const Case& c : cases) {
    std::vector<std::vector<double>> seq[2];
    LbfgsResult res[2];
    std::vector<double> xs[2];
    for (int which = 0; which < 2; ++which) {
      std::vector<double> x = c.x0;
      Objective fun = [&](const double* xx, double* g) {
        seq[which].push_back(std::vector<double>(xx, xx + c.n));
        return objective(c.kind, c.n, xx, g);
      };
      res[which] = which == 0 ? lbfgsb_minimize_loops(fun, x.data(), c.lo.data(), c.hi.data(), c.n, c.opt)
                              : lbfgsb_minimize(fun, x.data(), c.lo.data(), c.hi.data(), c.n, c.opt);
      xs[which] = x;
    }
    bool ok = seq[0].size() == seq[1].size() && res[0].nevals == res[1].nevals && res[0].iterations == res[1].iterations &&
              res[0].converged == res[1].converged && std::memcmp(&res[0].f, &res[1].f, 8) == 0 &&
              std::memcmp(xs[0].data(), xs[1].data(), 8 * c.n) == 0;
    for (size_t e = 0; ok && e < seq[0].size(); ++e) ok = std::memcmp(seq[0][e].data(), seq[1][e].data(), 8 * c.n) == 0;
    std::printf("%-28s %s: %zu / %zu evaluations, %d iterations, f = %.17g, converged %d\n", c.name, ok ? "same" : "DIFFERENT", seq[0].size(),
                seq[1].size(), res[1].iterations, res[1].f, (int)res[1].converged);
    if (!ok) ++bad;
  }
  return bad ? 1 : 0;
}
