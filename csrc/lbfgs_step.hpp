// lbfgs_step.hpp — the bounded L-BFGS of lbfgsb.hpp as a RESUMABLE state machine, usable on the host and on the device.
//
// lbfgsb_minimize() (lbfgsb.hpp) calls the objective from inside its loops, which needs the host in every evaluation.  For
// problems of at most 128 rows one evaluation is a single workgroup's work (small_eval_kernel), and the round trip through
// the host (graph launch + completion wake-up + three host threads on the runtime's lock: ~40 us) is as long as the
// evaluation.  Here the same algorithm is turned inside out: lbfgs_begin() returns the first point to evaluate,
// lbfgs_advance() takes that evaluation's (f, grad) and returns the next point or "finished".  The HOST path runs on it
// (lbfgsb_minimize is a loop around it; tests/cpp/test_lbfgs_step.cpp replays the old loop form's iterate sequence bit for bit).
// The persistent kernel (small_fit_kernel) shares the state layout (LbfgsState) and the method, but NOT this code: one thread
// walking it took 70 us per evaluation, so wave 0 runs a wave-wide transcription (kernels.hip: wl_begin_iteration / wl_advance --
// lane = dimension, dot products as DPP reductions, i.e. sums in another order, and the device's exp for theta -> parameters).
// That transcription is pinned against this file on the GPU: tests/test_gpu_fit.py replays a device run's trace
// (theta_i, f_i, g_i) through lbfgs_advance on the host and checks every next point it asks for.
//
// Same method, same constants, same evaluation counting as lbfgsb.hpp: projected L-BFGS (src/util/gradmin.rs:35-60 calls
// NLopt's bounded L-BFGS with maxeval = 150) -- active set from the sign of the gradient at the bounds, two-loop recursion
// on the free variables, projected backtracking (Armijo) line search with safeguarded quadratic interpolation.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define LBFGS_HD __host__ __device__
#else
#define LBFGS_HD
#endif

namespace hbegp {

constexpr int LBFGS_MAXN = 66;  // MAXP: noise, amplitude, up to 64 length scales
constexpr int LBFGS_MAXM = 10;  // history pairs kept

struct LbfgsState {
  // configuration
  int n, maxeval, m;
  double pgtol, ftol;
  int fixed_work;
  double lo[LBFGS_MAXN], hi[LBFGS_MAXN];
  // progress
  int phase;       // 0: first evaluation pending, 1: line-search evaluation pending, 2: burning evaluations, 3: finished
  int nevals, iterations, converged;
  int hcount;      // history pairs in use (oldest first)
  double f, step, gs;
  double x[LBFGS_MAXN], g[LBFGS_MAXN], xn[LBFGS_MAXN], d[LBFGS_MAXN];
  double S[LBFGS_MAXM][LBFGS_MAXN], Y[LBFGS_MAXM][LBFGS_MAXN], rho[LBFGS_MAXM];
};

namespace lbfgs_detail {
LBFGS_HD inline double dmin(double a, double b) { return a < b ? a : b; }
LBFGS_HD inline double dmax(double a, double b) { return a > b ? a : b; }
LBFGS_HD inline bool finite(double v) { return v - v == 0.0; }

// Proposes the next line-search point xn = clip(x + step d); false: the step no longer changes x.
LBFGS_HD inline bool propose(LbfgsState& s) {
  const int n = s.n;
  double gs = 0, moved = 0;
  for (int i = 0; i < n; ++i) {
    double v = s.x[i] + s.step * s.d[i];
    v = dmin(dmax(v, s.lo[i]), s.hi[i]);
    s.xn[i] = v;
    gs += s.g[i] * (v - s.x[i]);
    moved = dmax(moved, fabs(v - s.x[i]));
  }
  s.gs = gs;
  return moved != 0.0;
}

// Start of an outer iteration at (x, f, g): convergence test, search direction, first trial point.
// Returns the next phase: 1 (xn is to be evaluated) or 2 (go on to the burn / finish).
LBFGS_HD inline int begin_iteration(LbfgsState& s) {
  const int n = s.n;
  for (;;) {
    if (s.nevals >= s.maxeval) return 2;
    double pg[LBFGS_MAXN], q[LBFGS_MAXN], a[LBFGS_MAXM];
    double pgnorm = 0;
    for (int i = 0; i < n; ++i) {
      const bool at_lo = s.x[i] <= s.lo[i] && s.g[i] > 0, at_hi = s.x[i] >= s.hi[i] && s.g[i] < 0;
      pg[i] = (at_lo || at_hi) ? 0.0 : s.g[i];
      pgnorm = dmax(pgnorm, fabs(pg[i]));
    }
    if (pgnorm <= s.pgtol * dmax(1.0, fabs(s.f))) {
      s.converged = 1;
      return 2;
    }
    // two-loop recursion on the free variables
    for (int i = 0; i < n; ++i) q[i] = pg[i];
    const int hcount = s.hcount;
    for (int h = hcount - 1; h >= 0; --h) {
      double sq = 0;
      for (int i = 0; i < n; ++i) sq += s.S[h][i] * q[i];
      a[h] = s.rho[h] * sq;
      for (int i = 0; i < n; ++i) q[i] -= a[h] * s.Y[h][i];
    }
    double gamma = 1.0;
    if (hcount > 0) {
      double sy = 0, yy = 0;
      for (int i = 0; i < n; ++i) {
        sy += s.S[hcount - 1][i] * s.Y[hcount - 1][i];
        yy += s.Y[hcount - 1][i] * s.Y[hcount - 1][i];
      }
      if (yy > 0) gamma = sy / yy;
    }
    for (int i = 0; i < n; ++i) q[i] *= gamma;
    for (int h = 0; h < hcount; ++h) {
      double yq = 0;
      for (int i = 0; i < n; ++i) yq += s.Y[h][i] * q[i];
      const double b = s.rho[h] * yq;
      for (int i = 0; i < n; ++i) q[i] += s.S[h][i] * (a[h] - b);
    }
    double dg = 0;
    for (int i = 0; i < n; ++i) {
      s.d[i] = (pg[i] == 0.0) ? 0.0 : -q[i];
      dg += s.d[i] * s.g[i];
    }
    if (!(dg < 0)) {  // not a descent direction: fall back to projected steepest descent
      s.hcount = 0;
      dg = 0;
      for (int i = 0; i < n; ++i) {
        s.d[i] = -pg[i];
        dg += s.d[i] * s.g[i];
      }
    }
    s.step = 1.0;
    if (s.hcount == 0) {
      double dn = 0;
      for (int i = 0; i < n; ++i) dn += s.d[i] * s.d[i];
      s.step = dmin(1.0, 1.0 / sqrt(dmax(dn, 1e-300)));
    }
    if (propose(s)) return 1;
    // the very first step does not move x: the line search fails without an evaluation
    if (s.hcount > 0) {  // retry once from steepest descent with a clean history
      s.hcount = 0;
      continue;
    }
    return 2;
  }
}
}  // namespace lbfgs_detail

// Sets the run up at x0 (clipped into the box); the first point to evaluate is st.x.
LBFGS_HD inline void lbfgs_begin(LbfgsState& s, const double* x0, const double* lo, const double* hi, int n, int maxeval, int memory,
                                 double pgtol, double ftol, bool fixed_work) {
  s.n = n;
  s.maxeval = maxeval;
  s.m = memory < 1 ? 1 : (memory > LBFGS_MAXM ? LBFGS_MAXM : memory);
  s.pgtol = pgtol;
  s.ftol = ftol;
  s.fixed_work = fixed_work ? 1 : 0;
  s.phase = 0;
  s.nevals = s.iterations = s.converged = s.hcount = 0;
  s.f = INFINITY;
  s.step = 1.0;
  s.gs = 0;
  for (int i = 0; i < n; ++i) {
    s.lo[i] = lo[i];
    s.hi[i] = hi[i];
    s.x[i] = lbfgs_detail::dmin(lbfgs_detail::dmax(x0[i], lo[i]), hi[i]);
  }
}

// The point the caller has to evaluate next (valid while phase != 3).
LBFGS_HD inline const double* lbfgs_request(const LbfgsState& s) { return s.phase == 1 ? s.xn : s.x; }

// Feeds the evaluation of lbfgs_request(): f (may be +inf / NaN: a failed evaluation, grad then ignored) and grad.
// Returns true while another evaluation is wanted.
LBFGS_HD inline bool lbfgs_advance(LbfgsState& s, double fe, const double* ge) {
  using namespace lbfgs_detail;
  const int n = s.n;
  ++s.nevals;
  if (fe != fe) fe = INFINITY;
  int next = 2;
  if (s.phase == 0) {
    s.f = fe;
    for (int i = 0; i < n; ++i) s.g[i] = ge[i];
    // a failed start point: nothing to build a model on (lbfgsb.hpp reports +inf like NLopt's failure value)
    next = finite(fe) ? begin_iteration(s) : 2;
  } else if (s.phase == 1) {
    const double fn = fe;
    if (finite(fn) && fn <= s.f + 1e-4 * s.gs) {
      // accepted: curvature pair, move
      ++s.iterations;
      double sv[LBFGS_MAXN], yv[LBFGS_MAXN];
      double sy = 0, ss = 0, yy = 0;
      for (int i = 0; i < n; ++i) {
        sv[i] = s.xn[i] - s.x[i];
        yv[i] = ge[i] - s.g[i];
        sy += sv[i] * yv[i];
        ss += sv[i] * sv[i];
        yy += yv[i] * yv[i];
      }
      if (sy > 1e-10 * sqrt(ss * yy) && sy > 0) {
        if (s.hcount == s.m) {  // drop the oldest pair
          for (int h = 1; h < s.hcount; ++h) {
            for (int i = 0; i < n; ++i) {
              s.S[h - 1][i] = s.S[h][i];
              s.Y[h - 1][i] = s.Y[h][i];
            }
            s.rho[h - 1] = s.rho[h];
          }
          --s.hcount;
        }
        for (int i = 0; i < n; ++i) {
          s.S[s.hcount][i] = sv[i];
          s.Y[s.hcount][i] = yv[i];
        }
        s.rho[s.hcount] = 1.0 / sy;
        ++s.hcount;
      }
      const double fold = s.f;
      for (int i = 0; i < n; ++i) {
        s.x[i] = s.xn[i];
        s.g[i] = ge[i];
      }
      s.f = fn;
      if (fold - s.f <= s.ftol * dmax(1.0, fabs(s.f))) {
        s.converged = 1;  // tiny improvement: stop
        next = 2;
      } else {
        next = begin_iteration(s);
      }
    } else {
      // rejected: shrink the step (safeguarded quadratic interpolation through f, f'(0) = gs / step, fn)
      double nstep = 0.5 * s.step;
      if (finite(fn) && s.gs < 0) {
        const double denom = 2.0 * (fn - s.f - s.gs);
        if (denom > 0) nstep = dmin(dmax(-s.gs * s.step / denom, 0.1 * s.step), 0.5 * s.step);
      }
      s.step = nstep;
      bool again = !(s.step < 1e-20) && s.nevals < s.maxeval && propose(s);
      if (again) {
        next = 1;
      } else if (s.hcount > 0) {  // the line search failed: retry once from steepest descent with a clean history
        s.hcount = 0;
        next = s.nevals < s.maxeval ? begin_iteration(s) : 2;
      } else {
        next = 2;
      }
    }
  } else {
    next = 2;  // burning
  }
  if (next == 1) {
    s.phase = 1;
    return true;
  }
  // fixed-work mode: spend the remaining evaluations at the incumbent
  if (s.fixed_work && s.nevals < s.maxeval) {
    s.phase = 2;
    return true;
  }
  s.phase = 3;
  return false;
}

}  // namespace hbegp
