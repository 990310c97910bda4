"""The parity rules of the GPU tests, in one place.

f64: the north star's 1e-8 against the LAPACK oracle (oracle/gpr_oracle.py), relative to max(1, scale).  Where the oracle's own
digits end -- the optimiser visits corners of the box with cond(K) ~ 1e9..1e12, and a LAPACK solve is good to ~cond(K) eps --
the extended-precision referee (oracle/referee.py) supplies the truth and the rule becomes

    |gpu - truth| <= max(1e-8 * scale, 2 * |lapack - truth|)

i.e. the engine may be at most twice as far from the truth as the reference's own arithmetic is.  No multiple of cond(K) eps
appears anywhere.

f32 (--use-32): the same with 1e-4 and the f32 oracle (LAPACK spotrf/spotrs/spotri) as `lapack` -- plus one more term.  An f32
result starts from a kernel matrix whose entries carry about an ulp of f32 rounding, and at cond(K) ~ 1e6..1e7 that alone moves
the lml by 1e-2..1e-1: the error of ANY f32 evaluation there is a random draw, LAPACK's included (emulated on the CPU along a
fit's trajectory: LAPACK f32 0.185 / 0.002 / 0.09 at three neighbouring thetas, an exactly factored f32-rounded K 0.07 / 0.013 /
0.018), so "twice LAPACK's one draw" is not a bar.  The third term is the SCALE of that draw: the standard deviation of the
quantity's first-order change when every entry of K (and of dK/dtheta) is off by an independent relative error uniform in
+-1 ulp of f32 (`sigma32`, computed from f64 quantities).  Along the emulated trajectory LAPACK f32 sits at 0.1 - 7 sigma, an
exactly factored rounded K at 0.3 - 4 sigma; the allowance is 8 sigma.

    |gpu32 - truth| <= max(1e-4 * scale, 2 * |lapack32 - truth|, 8 * sigma32)

One case is outside every rule: where the reference's own arithmetic is further from the truth than the LARGEST entry of the
truth is from zero (|lapack - truth| >= max(1, max |truth|): not one correct digit -- f32 at a corner of the box with cond(K) beyond
1 / eps_f32; seen once, K^-1 of a captured f32 model: LAPACK off by 3.6e5 on entries of at most 3.0e4), twice one noise draw
against another noise draw decides nothing.  Such a comparison is counted (`n_nodigits`, shown in the summary), the engine's
values must be finite, and nothing else is asserted.
"""
import math

import numpy as np

from oracle import gpr_oracle as O
from oracle import referee as R

TOL64, TOL32 = 1e-8, 1e-4
N_SIGMA = 8.0  # f32 only: standard deviations of the first-order effect of +-1 ulp (f32) perturbations of K


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
        self.n_nodigits = 0     # comparisons where the reference's arithmetic has no correct digit (see the module text)

    def check(self, what, got, lapack, truth_fn, scale=None, sigma_fn=None):
        d = dev(got, lapack, scale)
        if d <= self.tol:
            self.n_plain += 1
            self.worst_plain = max(self.worst_plain, d)
            return
        truth = truth_fn()
        s = max(1.0, float(np.max(np.abs(truth))) if scale is None else float(scale))
        e_gpu = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - truth)))
        e_lap = float(np.max(np.abs(np.asarray(lapack, dtype=np.float64) - truth)))
        if e_lap >= s:  # (s = max(1, largest entry of the truth): the scale every bar here is relative to)
            self.n_nodigits += 1
            assert np.all(np.isfinite(np.asarray(got, dtype=np.float64))), f"{what}: not finite"
            return
        allowed = max(self.tol * s, 2.0 * e_lap)
        if sigma_fn is not None:
            allowed = max(allowed, N_SIGMA * float(np.max(sigma_fn())))
        self.n_refereed += 1
        self.worst_ratio = max(self.worst_ratio, e_gpu / allowed)
        assert e_gpu <= allowed, f"{what}: |gpu - truth| = {e_gpu:.3e} > max({self.tol:g} * {s:.3g}, 2 * |lapack - truth| = {2 * e_lap:.3e})"

    def summary(self):
        return (f"{self.n_plain} comparisons inside {self.tol:g} of the oracle (worst {self.worst_plain:.2e}), {self.n_refereed} refereed "
                f"(worst |gpu - truth| / allowance {self.worst_ratio:.2f})" +
                (f", {self.n_nodigits} where the reference's arithmetic has no correct digit (not judged)" if self.n_nodigits else ""))


def referee_for(X, y, theta, bounds, nu=2.5):
    noise, c, ell = clamped_params(theta, bounds)
    return R.Referee(np.asarray(X, dtype=np.float64), np.asarray(y, dtype=np.float64), noise, c, ell, nu)


def sigma32(X, y, theta, bounds, nu=2.5):
    """Standard deviation of the first-order change of lml, gradient, alpha and K^-1 when every entry of K and of dK/dtheta_j
    carries an independent relative error uniform in +-eps_f32 (std eps / sqrt 3) -- the scale of what storing the kernel matrix
    in f32 costs, whatever the algorithm.  From f64 quantities (used where cond(K) <= 1e8).  With W = a a^T - K^-1, V = K^-1:
      d lml = 1/2 tr(W dK);   d a = -V dK a;   d V = -V dK V;
      d g_j = 1/2 tr(C_j dK) + 1/2 tr(W dD_j),  D_j = dK/dtheta_j,  C_j = V D_j V - a (V D_j a)^T - (V D_j a) a^T."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    noise, c, ell = clamped_params(theta, bounds)
    u = float(np.finfo(np.float32).eps) / math.sqrt(3.0)
    kernel, dk = O.product_theta_grad(X, c, ell, nu)
    n = len(y)
    K = kernel + noise * np.eye(n)
    V = np.linalg.inv(K)
    a = V @ y
    W = np.outer(a, a) - V
    K2, V2 = K * K, V * V
    s_alpha = u * np.sqrt(V2 @ ((K * a[None, :]) ** 2).sum(axis=1))
    s_kinv = u * np.sqrt(V2 @ K2 @ V2)
    grads = []
    for D in [noise * np.eye(n)] + [dk[:, :, j] for j in range(dk.shape[2])]:
        VDa = V @ (D @ a)
        C = V @ D @ V - np.outer(a, VDa) - np.outer(VDa, a)
        grads.append(0.5 * u * math.sqrt(float(((C * K) ** 2).sum() + ((W * D) ** 2).sum())))
    return {"lml": 0.5 * u * math.sqrt(float(((W * K) ** 2).sum())), "grad": np.array(grads), "alpha": s_alpha, "kinv": s_kinv}
