"""The parity rules of the GPU tests, in one place.

f64: the north star's 1e-8 against the LAPACK oracle (oracle/gpr_oracle.py), relative to max(1, scale).  Where the oracle's own
digits end -- the optimiser visits corners of the box with cond(K) ~ 1e9..1e12, and a LAPACK solve is good to ~cond(K) eps --
the extended-precision referee (oracle/referee.py) supplies the truth and the rule becomes

    |gpu - truth| <= max(1e-8 * scale, 2 * |lapack - truth|)

i.e. the engine may be at most twice as far from the truth as the reference's own arithmetic is.  No multiple of cond(K) eps
appears anywhere.  f32: the same with 1e-4 and the f32 oracle (LAPACK spotrf/spotrs/spotri) as `lapack`.
"""
import math

import numpy as np

from oracle import gpr_oracle as O
from oracle import referee as R

TOL64, TOL32 = 1e-8, 1e-4


def clamped_params(theta, bounds):
    """fit.rs:94-96: noise = exp(theta_0) unclamped, kernel parameters clamped into their bounds after exp."""
    noise = math.exp(theta[0])
    c = O.clamp(math.exp(theta[1]), *bounds[1])
    ell = np.array([O.clamp(math.exp(t), lo, hi) for t, (lo, hi) in zip(theta[2:], bounds[2:])])
    return noise, c, ell


def dev(got, want, scale=None):
    """max |got - want| / max(1, scale of want)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    s = float(np.max(np.abs(want))) if scale is None else float(scale)
    return float(np.max(np.abs(got - want))) / max(1.0, s)


class Judge:
    """Collects what was compared and how; `check` applies the rule above.  `truth_fn` is called only when the plain bar fails."""

    def __init__(self, tol=TOL64):
        self.tol = tol
        self.n_plain = 0
        self.n_refereed = 0
        self.worst_plain = 0.0
        self.worst_ratio = 0.0  # |gpu - truth| / max(tol * scale, 2 |lapack - truth|) over the refereed comparisons

    def check(self, what, got, lapack, truth_fn, scale=None):
        d = dev(got, lapack, scale)
        if d <= self.tol:
            self.n_plain += 1
            self.worst_plain = max(self.worst_plain, d)
            return
        truth = truth_fn()
        s = max(1.0, float(np.max(np.abs(truth))) if scale is None else float(scale))
        e_gpu = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - truth)))
        e_lap = float(np.max(np.abs(np.asarray(lapack, dtype=np.float64) - truth)))
        allowed = max(self.tol * s, 2.0 * e_lap)
        self.n_refereed += 1
        self.worst_ratio = max(self.worst_ratio, e_gpu / allowed)
        assert e_gpu <= allowed, f"{what}: |gpu - truth| = {e_gpu:.3e} > max({self.tol:g} * {s:.3g}, 2 * |lapack - truth| = {2 * e_lap:.3e})"

    def summary(self):
        return (f"{self.n_plain} comparisons inside {self.tol:g} of the oracle (worst {self.worst_plain:.2e}), {self.n_refereed} refereed "
                f"(worst |gpu - truth| / allowance {self.worst_ratio:.2f})")


def referee_for(X, y, theta, bounds, nu=2.5):
    noise, c, ell = clamped_params(theta, bounds)
    return R.Referee(np.asarray(X, dtype=np.float64), np.asarray(y, dtype=np.float64), noise, c, ell, nu)
