"""GPU, BASELINE.json's full sizes: size-independent properties of one evaluation / model (the oracle would take minutes
at these sizes).  K comes from the CPU oracle's kernel (independent of the device assembly kernel)."""
import math
import os

import numpy as np
import pytest

from hbetune_rs_amd import gpr, synth
from oracle import gpr_oracle as O

pytestmark = pytest.mark.gpu


def _theta_parts(theta):
    return math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])


@pytest.mark.parametrize("cfg", ["M", "C3", "C4", "C5"])
def test_full_size_properties(cfg):
    w = synth.make_workload(cfg)
    X, y, theta = w["X"], w["y"], w["theta"].copy()
    f32 = X.dtype == np.float32
    if f32:
        theta[0] = theta[1] + math.log(0.5)  # keep cond(K) moderate for f32 (see test_f32_path_himmelblau)
    s2, c, ell = _theta_parts(theta)
    n, d = X.shape
    tol = 2e-3 if f32 else 1e-8
    prob = gpr.Problem(X, y)
    res = prob.lml_with_gradient(theta)
    assert res is not None
    lml, grad = res
    alpha, _, ldiag = prob.results(want_kinv=False)
    assert np.all(np.isfinite(alpha)) and np.all(ldiag > 0)

    # (1) alpha solves (K + s2 I) alpha = y: residual relative to |y|, K from the CPU oracle in f64
    K = O.product_kernel(X.astype(np.float64), X.astype(np.float64), c, ell, 2.5)
    K[np.diag_indices(n)] += s2
    r = K @ alpha.astype(np.float64) - y.astype(np.float64)
    assert np.linalg.norm(r) <= tol * np.linalg.norm(y) * 10

    # (2) the device kernel matrix equals the oracle's
    Kd = prob.kernel_matrix(theta)
    np.testing.assert_allclose(Kd[::97, ::89], K[::97, ::89], rtol=1e-5 if f32 else 1e-12, atol=1e-6 if f32 else 1e-14)

    # (3) lml identity: -1/2 y^T alpha - sum log L_ii - n/2 log 2 pi, with log det from an independent f64 Cholesky
    sign, logdet = np.linalg.slogdet(K)
    want_lml = -0.5 * float(y.astype(np.float64) @ alpha.astype(np.float64)) - 0.5 * logdet - n / 2 * math.log(2 * math.pi)
    assert abs(lml - want_lml) <= (1e-3 if f32 else 1e-9) * abs(want_lml)

    # (4) gradient = directional derivative of lml (central difference along a fixed direction)
    rng = np.random.default_rng(3)
    u = rng.standard_normal(d + 2)
    u /= np.linalg.norm(u)
    h = 1e-3 if f32 else 1e-5
    lp = prob.lml_with_gradient(theta + h * u, want_grad=False)[0]
    lm = prob.lml_with_gradient(theta - h * u, want_grad=False)[0]
    fd = (lp - lm) / (2 * h)
    assert abs(fd - grad @ u) <= (5e-2 if f32 else 1e-5) * max(1.0, np.abs(grad).max())

    # (5) predict at training points: mean_i = y_i - s2 alpha_i exactly (K* = K - s2 I), variance in [0, c + 1e-5]
    fk = gpr.FittedKernel.extend(X, y, theta)
    idx = np.arange(0, n, max(1, n // 512))
    mean, var, _ = fk.predict(X[idx])
    np.testing.assert_allclose(mean.astype(np.float64), y[idx].astype(np.float64) - s2 * alpha[idx].astype(np.float64), rtol=0,
                               atol=(1e-2 if f32 else 1e-8) * max(1.0, np.abs(y).max()))
    assert np.all(var >= 0) and np.all(var <= c + 1e-4)
    # (6) K^-1 is an inverse: K (K^-1 v) = v for the model's matrix, checked on its diagonal identity
    #     var_i = 1e-5 + s2 - s2^2 Kinv_ii  (from k*_i = K_i - s2 e_i)
    if n <= 4096:
        _, kinv = fk.arrays()
        want_var = 1e-5 + s2 - s2 * s2 * np.diag(kinv)[idx].astype(np.float64)
        np.testing.assert_allclose(var.astype(np.float64), np.maximum(want_var, 0), rtol=0, atol=(2e-2 if f32 else 1e-8) * c)
        v = rng.standard_normal(n)
        back = K @ (kinv.astype(np.float64) @ v)
        assert np.linalg.norm(back - v) <= (5e-2 if f32 else 1e-8) * np.linalg.norm(v) * 10


@pytest.mark.parametrize("cfg", ["M", "C3", "C4"])
def test_full_size_against_the_oracle(cfg):
    """The f64 BASELINE configurations at their FULL size next to the oracle (lml.rs:29-79, predict.rs:7-52 restated with the
    reference's own LAPACK calls), through the default path -- the right-looking task queue, the code that produces the
    headline number: lml, gradient, alpha, the lower triangle of K^-1, and mean / variance at 64 candidates, all at 1e-8
    (relative to max(1, scale); the variance relative to the amplitude).  M: n=4096 d=8; C3: n=4096 d=16; C4: n=8192 d=2
    (logarithmic projection).  Host time of the oracle: ~6 / ~12 / ~40 s."""
    w = synth.make_workload(cfg)
    X, y, theta = w["X"], w["y"], w["theta"]
    assert X.dtype == np.float64
    s2, c, ell = _theta_parts(theta)
    n, d = X.shape
    ref = O.lml_with_gradient(X, y, s2, c, ell, 2.5)
    prob = gpr.Problem(X, y)
    lml, grad = prob.lml_with_gradient(theta)
    alpha, kinv, ldiag = prob.results()
    prob.close()
    tol = 1e-8
    cond_lb = float((ldiag.max() / ldiag.min()) ** 2)  # a lower bound on cond(K) from the factor's diagonal
    dev = {
        "lml": abs(lml - ref["lml"]) / max(1.0, abs(ref["lml"])),
        "grad": float(np.max(np.abs(grad - ref["grad"])) / max(1.0, np.abs(ref["grad"]).max())),
        "alpha": float(np.max(np.abs(alpha - ref["alpha"])) / max(1.0, np.abs(ref["alpha"]).max())),
        "k_inv": float(np.max(np.abs(np.tril(kinv) - np.tril(ref["k_inv"]))) / max(1.0, np.abs(ref["k_inv"]).max())),
    }
    del kinv
    fk = gpr.FittedKernel.extend(X, y, theta)
    Xs = synth.candidates(cfg, 64, d)
    mean, var, _ = fk.predict(Xs)
    fk.release()
    rm, rv, _ = O.predict(Xs, X, ref["alpha"], ref["k_inv"], c, ell, 2.5)
    dev["mean"] = float(np.max(np.abs(mean - rm)) / max(1.0, np.abs(rm).max()))
    dev["var"] = float(np.max(np.abs(var - rv)) / c)
    print(f"{cfg} n={n} d={d}: deviation from the oracle / scale: {dev}; cond(K) >= {cond_lb:.2e}")
    assert all(v <= tol for v in dev.values()), (dev, cond_lb)


def test_c3_eight_restart_fit_short():
    # C3: 8 optimiser runs (1 + 7 restarts) on one GPU, shortened to 6 evaluations per run; the capture must be the
    # arg-max over all 48 evaluations and the model must predict with that theta
    w = synth.make_workload("C3")
    starts = synth.restart_points("C3", w["lo"], w["hi"], 7)
    fk = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], starts, maxeval=6, trace=True)
    tr = fk.trace
    assert len(tr["lml"]) <= 48 and set(tr["run"].tolist()) == set(range(8))
    assert fk.lml == tr["lml"].max()
    mean, var, _ = fk.predict(w["X"][:16])
    assert np.all(np.isfinite(mean)) and np.all(var >= 0)


@pytest.mark.parametrize("noise_over_amplitude", [1e-2, 0.5])
def test_c5_full_size_f32_against_both_oracles(noise_over_amplitude):
    """BASELINE config C5 at its full n = 2048 (`--use-32`, main.rs:240-244): the GPU's f32 results against f64 truth (the
    reference's arithmetic with A = f64 on the same f32 inputs) AND against the f32 oracle (the reference's own f32
    arithmetic, LAPACK spotrf/spotrs/spotri).  Pass iff |gpu32 - f64| <= max(1e-4 * scale, 2 * |lapack32 - f64|) for lml,
    gradient, alpha, K^-1, mean and variance -- at SURVEY 8(d)'s theta (sigma^2 = 1e-2 c) and at sigma^2 = 0.5 c."""
    w = synth.make_workload("C5")
    X, y = w["X"], w["y"]
    assert X.dtype == np.float32 and X.shape == (2048, 2)
    theta = w["theta"].copy()
    theta[0] = theta[1] + math.log(noise_over_amplitude)
    s2, c, ell = _theta_parts(theta)
    X64, y64 = X.astype(np.float64), y.astype(np.float64)
    r64 = O.lml_with_gradient(X64, y64, s2, c, ell, 2.5)
    r32 = O.lml_with_gradient(X, y, s2, c, ell, 2.5)
    cond = float(np.linalg.cond(r64["kernel_matrix"]))
    prob = gpr.Problem(X, y)
    lml, grad = prob.lml_with_gradient(theta)
    alpha, kinv, _ = prob.results()
    fk = gpr.FittedKernel.extend(X, y, theta)
    Xs = synth.candidates("C5", 128, 2).astype(np.float32)
    mean, var, _ = fk.predict(Xs)
    m64, v64, _ = O.predict(Xs.astype(np.float64), X64, r64["alpha"], r64["k_inv"], c, ell, 2.5)
    m32, v32, _ = O.predict(Xs, X, r32["alpha"], r32["k_inv"], c, ell, 2.5)

    def check(name, got, truth, lapack, scale):
        dev = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - truth)))
        ref = float(np.max(np.abs(np.asarray(lapack, dtype=np.float64) - truth)))
        assert dev <= max(1e-4 * scale, 2 * ref), f"{name}: gpu32 off by {dev:.3e}, lapack32 by {ref:.3e}, scale {scale:.3e}, cond(K) {cond:.2e}"
        return dev / scale, ref / scale

    report = {
        "lml": check("lml", lml, r64["lml"], r32["lml"], abs(r64["lml"])),
        "grad": check("grad", grad, r64["grad"], r32["grad"], np.abs(r64["grad"]).max()),
        "alpha": check("alpha", alpha, r64["alpha"], r32["alpha"], np.abs(r64["alpha"]).max()),
        "k_inv": check("k_inv", kinv, r64["k_inv"], r32["k_inv"], np.abs(r64["k_inv"]).max()),
        "mean": check("mean", mean, m64, m32, max(1.0, np.abs(m64).max())),
        "var": check("var", var, np.maximum(v64, 0), np.maximum(v32, 0), c),
    }
    print(f"C5 n=2048 f32, sigma^2/c={noise_over_amplitude}, cond(K)={cond:.3e}: (gpu32, lapack32) deviation from f64 / scale:", report)
    # with the chunked fp64 totals of the f32 tile GEMM the engine also meets the plain 1e-4 bar here (cond(K) up to 7e4) -- on
    # the path a user gets (the right-looking task queue at this size).  When the environment forces the recursion order
    # (HBEGP_DAG=0 / HBEGP_DAG_RL=0: panel solves through the explicit inverse of the whole left half) K^-1 sits at 1.2e-4 at
    # cond(K) = 7e4 -- inside the rule above (LAPACK f32: 1.7e-4), outside the plain bar -- so the plain bar is asserted for
    # the default configuration only (tools/gpu_matrix.sh runs the suite under the forced configurations).
    if os.environ.get("HBEGP_DAG") != "0" and os.environ.get("HBEGP_DAG_RL") != "0":
        assert all(v[0] <= 1e-4 for v in report.values()), report
