"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the sklearn golden vectors.

Tolerances (BASELINE.json north_star): f64 1e-8, f32 1e-4 — applied to the user-visible outputs (mean, variance, alpha)
absolutely and to lml/gradient relative to their scale; every fixture records cond(K).
"""
import glob
import math
import os

import numpy as np
import pytest

from hbetune_rs_amd import gpr, synth
from oracle import gpr_oracle as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))
F64_TOL = 1e-8
F32_TOL = 1e-4


def split_theta(theta):
    return math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_lml_grad_predict_match_golden_f64(path):
    g = np.load(path)
    X, y, theta, nu = g["X"], g["y"], g["theta"], float(g["nu"])
    prob = gpr.Problem(X, y, nu=nu)
    K = prob.kernel_matrix(theta)
    np.testing.assert_allclose(K[:4], g["K_rows"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(np.diag(K), g["K_diag"], rtol=1e-12)
    assert abs(K.sum() - float(g["K_sum"])) <= 1e-10 * abs(float(g["K_sum"]))
    res = prob.lml_with_gradient(theta)
    assert res is not None
    lml, grad = res
    assert abs(lml - float(g["lml"])) <= F64_TOL * max(1.0, abs(float(g["lml"])))
    np.testing.assert_allclose(grad, g["grad"], rtol=0, atol=F64_TOL * max(1.0, np.abs(g["grad"]).max()))
    alpha, kinv, ldiag = prob.results()
    np.testing.assert_allclose(alpha, g["alpha"], rtol=0, atol=F64_TOL * max(1.0, np.abs(g["alpha"]).max()))
    # K^-1 against the oracle (LAPACK potri)
    s2, c, ell = split_theta(theta)
    ref = O.lml_with_gradient(X, y, s2, c, ell, nu)
    np.testing.assert_allclose(kinv, ref["k_inv"], rtol=0, atol=F64_TOL * np.abs(ref["k_inv"]).max())
    np.testing.assert_allclose(ldiag, np.diag(ref["chol"]), rtol=1e-10)
    # model + predict
    fk = gpr.FittedKernel.extend(X, y, theta, nu=nu)
    mean, var, n_warn = fk.predict(g["Xs"])
    np.testing.assert_allclose(mean, g["mean"], rtol=0, atol=F64_TOL * max(1.0, np.abs(g["mean"]).max()))
    np.testing.assert_allclose(var, g["var"], rtol=0, atol=F64_TOL * c)
    assert n_warn == 0
    a2, k2 = fk.arrays()
    np.testing.assert_allclose(k2, ref["k_inv"], rtol=0, atol=F64_TOL * np.abs(ref["k_inv"]).max())
    assert np.array_equal(k2, k2.T)  # invc_into() hands back the full symmetric matrix
    assert abs(fk.lml - float(g["lml"])) <= F64_TOL * max(1.0, abs(float(g["lml"])))


@pytest.mark.parametrize("cfg,n", [("C2", 1024), ("C1", 64), ("M", 640), ("C3", 384), ("C4", 1100)])
def test_against_oracle_mid_sizes_f64(cfg, n):
    w = synth.make_workload(cfg, n=n)
    X, y, theta = w["X"], w["y"], w["theta"]
    s2, c, ell = split_theta(theta)
    ref = O.lml_with_gradient(X, y, s2, c, ell, 2.5)
    prob = gpr.Problem(X, y, nu=2.5)
    lml, grad = prob.lml_with_gradient(theta)
    assert abs(lml - ref["lml"]) <= F64_TOL * max(1.0, abs(ref["lml"]))
    np.testing.assert_allclose(grad, ref["grad"], rtol=0, atol=F64_TOL * max(1.0, np.abs(ref["grad"]).max()))
    alpha, kinv, _ = prob.results()
    np.testing.assert_allclose(alpha, ref["alpha"], rtol=0, atol=F64_TOL * max(1.0, np.abs(ref["alpha"]).max()))
    np.testing.assert_allclose(kinv, ref["k_inv"], rtol=0, atol=F64_TOL * np.abs(ref["k_inv"]).max())
    Xs = synth.candidates(cfg, 200, w["d"])
    fk = gpr.FittedKernel.extend(X, y, theta)
    mean, var, _ = fk.predict(Xs)
    rmean, rvar, _ = O.predict(Xs, X, ref["alpha"], ref["k_inv"], c, ell, 2.5)
    np.testing.assert_allclose(mean, rmean, rtol=0, atol=F64_TOL * max(1.0, np.abs(rmean).max()))
    np.testing.assert_allclose(var, rvar, rtol=0, atol=F64_TOL * c)


def test_recycled_workspaces_keep_triangular_operands_clean():
    """The tile GEMM does not mask triangular operands: the strict upper triangle of the L^-1 workspace has to be zero in
    memory.  Workspaces come from a pool that also recycles full symmetric K^-1 buffers, so run several problems of one
    size back to back (each leaves dirty buffers behind) and check every one against the oracle."""
    n = 384
    for rep, cfg in enumerate(["C3", "M", "C2", "C3"]):
        w = synth.make_workload(cfg, n=n)
        X, y, theta = w["X"], w["y"], w["theta"]
        s2, c, ell = split_theta(theta)
        ref = O.lml_with_gradient(X, y, s2, c, ell, 2.5)
        fk = gpr.FittedKernel.extend(X, y, theta)          # leaves a model (K^-1, L^-1 copies) in the pool when dropped
        prob = gpr.Problem(X, y, nu=2.5, n_slots=2)
        for slot in (0, 1):
            lml, grad = prob.lml_with_gradient(theta, slot=slot)
            assert abs(lml - ref["lml"]) <= F64_TOL * max(1.0, abs(ref["lml"])), (rep, slot)
            np.testing.assert_allclose(grad, ref["grad"], rtol=0, atol=F64_TOL * max(1.0, np.abs(ref["grad"]).max()))
            _, kinv, _ = prob.results(slot=slot)
            np.testing.assert_allclose(kinv, ref["k_inv"], rtol=0, atol=F64_TOL * np.abs(ref["k_inv"]).max())
        Xs = synth.candidates(cfg, 64, w["d"])
        _, var, _ = fk.predict(Xs)
        _, rvar, _ = O.predict(Xs, X, ref["alpha"], ref["k_inv"], c, ell, 2.5)
        np.testing.assert_allclose(var, rvar, rtol=0, atol=F64_TOL * c)
        del prob, fk


@pytest.mark.parametrize("noise_rel,ell_scale", [(1e-4, 4.0), (1e-8, 8.0)])
def test_ill_conditioned_kernel_matrix_stays_inside_the_bar(noise_rel, ell_scale):
    """Long length scales and tiny noise: cond(K) = 2e5 / 8.5e6.  The engine solves the triangular systems with explicit
    inverses; it must not lose digits against LAPACK there (tools/illcond_accuracy.py: both are 1e-10 from the truth)."""
    w = synth.make_workload("C2", n=384)
    X, y, theta = w["X"], w["y"], w["theta"].copy()
    theta[0] = theta[1] + math.log(noise_rel)
    theta[2:] += math.log(ell_scale)
    s2, c, ell = split_theta(theta)
    ref = O.lml_with_gradient(X, y, s2, c, ell, 2.5)
    prob = gpr.Problem(X, y, nu=2.5)
    lml, grad = prob.lml_with_gradient(theta)
    assert abs(lml - ref["lml"]) <= F64_TOL * max(1.0, abs(ref["lml"]))
    np.testing.assert_allclose(grad, ref["grad"], rtol=0, atol=F64_TOL * max(1.0, np.abs(ref["grad"]).max()))
    alpha, kinv, _ = prob.results()
    np.testing.assert_allclose(alpha, ref["alpha"], rtol=0, atol=F64_TOL * max(1.0, np.abs(ref["alpha"]).max()))
    np.testing.assert_allclose(kinv, ref["k_inv"], rtol=0, atol=F64_TOL * np.abs(ref["k_inv"]).max())


@pytest.mark.parametrize("nu", [0.5, 1.5, 2.5, float("inf")])  # inf = squared exponential (extension; oracle pinned on sklearn RBF goldens)
def test_all_matern_orders(nu):
    w = synth.make_workload("C1")
    s2, c, ell = split_theta(w["theta"])
    ref = O.lml_with_gradient(w["X"], w["y"], s2, c, ell, nu)
    prob = gpr.Problem(w["X"], w["y"], nu=nu)
    lml, grad = prob.lml_with_gradient(w["theta"])
    assert abs(lml - ref["lml"]) <= F64_TOL * max(1.0, abs(ref["lml"]))
    np.testing.assert_allclose(grad, ref["grad"], rtol=0, atol=F64_TOL * max(1.0, np.abs(ref["grad"]).max()))


def test_f32_path_himmelblau():
    # C5 (f32, --use-32) at a size the oracle finishes quickly: the plain 1e-4 bar of the north star against the f64 oracle
    # (reference arithmetic with A = f64 on the same f32 inputs), no multiples.  The f32 oracle (the reference's own f32
    # arithmetic: LAPACK spotrf/spotrs/spotri) is shown beside it: the engine must not be further from f64 than that.
    w = synth.make_workload("C5", n=512)
    assert w["X"].dtype == np.float32
    theta = w["theta"].copy()  # sigma^2 = 1e-2 c
    s2, c, ell = split_theta(theta)
    X64, y64 = w["X"].astype(np.float64), w["y"].astype(np.float64)
    ref = O.lml_with_gradient(X64, y64, s2, c, ell, 2.5)
    r32 = O.lml_with_gradient(w["X"], w["y"], s2, c, ell, 2.5)
    prob = gpr.Problem(w["X"], w["y"], nu=2.5)
    res = prob.lml_with_gradient(theta)
    assert res is not None
    lml, grad = res
    alpha, kinv, _ = prob.results()
    gscale = max(1.0, np.abs(ref["grad"]).max())
    assert abs(lml - ref["lml"]) <= F32_TOL * abs(ref["lml"])
    assert np.abs(grad - ref["grad"]).max() <= F32_TOL * gscale
    assert np.abs(alpha - ref["alpha"]).max() <= F32_TOL * np.abs(ref["alpha"]).max()
    assert np.abs(kinv - ref["k_inv"]).max() <= F32_TOL * np.abs(ref["k_inv"]).max()
    assert np.abs(grad - ref["grad"]).max() <= 2 * np.abs(r32["grad"] - ref["grad"]).max() + 1e-6 * gscale
    fk = gpr.FittedKernel.extend(w["X"], w["y"], theta)
    Xs = synth.candidates("C5", 64, 2).astype(np.float32)
    mean, var, _ = fk.predict(Xs)
    assert mean.dtype == np.float32 and var.dtype == np.float32
    rmean, rvar, _ = O.predict(Xs.astype(np.float64), X64, ref["alpha"], ref["k_inv"], c, ell, 2.5)
    assert np.abs(mean - rmean).max() <= F32_TOL * max(1.0, np.abs(rmean).max())
    assert np.abs(var - np.maximum(rvar, 0)).max() <= F32_TOL * c
    # the f32 fit and the incremental extend against the f64 oracle as well.  The noise floor is raised to 1e-2 c for the
    # fit: with the default 1e-5 the optimiser walks to cond(K) ~ 1e8, where no f32 arithmetic means anything (the reference
    # warns about exactly that, README.md:56-59)
    lo_fit = w["lo"].copy()
    lo_fit[0] = 1e-2 * c
    starts = synth.restart_points("C5", lo_fit, w["hi"], 1)
    fit = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], lo_fit, w["hi"], starts, maxeval=30)
    f, _, fres = O.objective(fit.theta, X64, y64, 2.5, list(zip(lo_fit, w["hi"])))
    assert abs(fit.lml + f) <= F32_TOL * max(1.0, abs(f))
    np.testing.assert_allclose(fit.arrays(False)[0], fres["alpha"], rtol=0, atol=5 * F32_TOL * max(1.0, np.abs(fres["alpha"]).max()))
    prior = gpr.FittedKernel.extend(w["X"][:400], w["y"][:400], theta)
    inc = prior.extend_with(w["X"], w["y"])
    assert inc.incremental and abs(inc.lml - ref["lml"]) <= F32_TOL * abs(ref["lml"])
    np.testing.assert_allclose(inc.arrays(False)[0], ref["alpha"], rtol=0, atol=F32_TOL * np.abs(ref["alpha"]).max())


def test_not_positive_definite_contract():
    # duplicate rows with vanishing noise: reference returns None -> +inf / zero gradient (lml.rs:47-50, fit.rs:105-112)
    X = np.array([[0.1, 0.2], [0.1, 0.2], [0.5, 0.5], [0.9, 0.1]])
    y = np.array([1.0, 2.0, 3.0, 0.5])
    theta = np.array([math.log(1e-300), 0.0, 0.0, 0.0])
    prob = gpr.Problem(X, y)
    assert prob.lml_with_gradient(theta) is None
    with pytest.raises(gpr.HbegpError) as e:
        gpr.FittedKernel.extend(X, y, theta)
    assert e.value.code == gpr.NOT_PD
    # and the slot recovers for the next theta
    ok = prob.lml_with_gradient(np.array([math.log(1e-2), 0.0, 0.0, 0.0]))
    assert ok is not None and np.isfinite(ok[0])


def test_bitwise_reproducible():
    w = synth.make_workload("C2", n=512)
    prob = gpr.Problem(w["X"], w["y"])
    a = prob.lml_with_gradient(w["theta"])
    b = prob.lml_with_gradient(w["theta"])
    assert a[0] == b[0] and np.array_equal(a[1], b[1])


def test_clamped_theta_and_bounds():
    # kernel parameters are clamped into their bounds after exp(), the noise is not (fit.rs:94-96)
    w = synth.make_workload("C1")
    theta = w["theta"].copy()
    theta[2] = math.log(1e4)  # beyond the 1e3 upper bound
    prob = gpr.Problem(w["X"], w["y"])
    got = prob.lml_with_gradient(theta, w["lo"], w["hi"])
    theta_c = theta.copy()
    theta_c[2] = math.log(1e3)
    want = prob.lml_with_gradient(theta_c)
    # exp(ln(1e3)) is 1e3 only to an ulp, so compare to rounding rather than bitwise
    assert abs(got[0] - want[0]) <= 1e-10 * abs(want[0])
    np.testing.assert_allclose(got[1], want[1], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("n,d", [(1, 1), (2, 3), (17, 1), (129, 2), (130, 64), (300, 5)])
def test_edge_shapes(n, d):
    # ragged sizes: single observation, n just above a 128 block, the maximum feature count (MAXD = 64)
    rng = np.random.default_rng(n * 100 + d)
    X = rng.random((n, d))
    y = np.sin(X.sum(axis=1) * 3) + 0.05
    theta = np.concatenate([[math.log(0.05), math.log(1.3)], np.log(0.4 + 0.1 * rng.random(d))])
    s2, c, ell = split_theta(theta)
    ref = O.lml_with_gradient(X, y, s2, c, ell, 2.5)
    prob = gpr.Problem(X, y)
    lml, grad = prob.lml_with_gradient(theta)
    assert abs(lml - ref["lml"]) <= F64_TOL * max(1.0, abs(ref["lml"]))
    np.testing.assert_allclose(grad, ref["grad"], rtol=0, atol=F64_TOL * max(1.0, np.abs(ref["grad"]).max()))
    fk = gpr.FittedKernel.extend(X, y, theta)
    Xs = rng.random((5, d))
    mean, var, _ = fk.predict(Xs)
    rmean, rvar, _ = O.predict(Xs, X, ref["alpha"], ref["k_inv"], c, ell, 2.5)
    np.testing.assert_allclose(mean, rmean, rtol=0, atol=F64_TOL * max(1.0, np.abs(rmean).max()))
    np.testing.assert_allclose(var, rvar, rtol=0, atol=F64_TOL * c)
    # empty candidate batch
    m0, v0, _ = fk.predict(np.zeros((0, d)))
    assert m0.shape == (0,) and v0.shape == (0,)


def test_invalid_arguments_are_rejected():
    X = np.random.default_rng(0).random((8, 2))
    y = np.ones(8)
    with pytest.raises(gpr.HbegpError) as e:
        gpr.Problem(X, y, nu=2.0)  # matern_kernel.rs:79: unimplemented!
    assert e.value.code == -1
    with pytest.raises(gpr.HbegpError):
        gpr.Problem(np.random.default_rng(0).random((8, 65)), y)  # d > MAXD
    fk = gpr.FittedKernel.extend(X, y + np.arange(8) * 0.1, np.array([math.log(0.1), 0.0, 0.0, 0.0]))
    with pytest.raises(AssertionError):
        fk.predict(np.zeros((3, 5)))  # wrong feature count


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m", [1, 2, 5, 16])
def test_handful_of_candidates_path_matches_oracle_and_batched_path(m, dtype):
    # predict.rs:7-52 for m <= 16 runs a dedicated path (L^-1 read once, no 128-row padding): the caller's scalar
    # predict_mean / predict_mean_ei / predict_confidence_bound loops (acquisition.rs:46-64, minimize.rs:656-714)
    w = synth.make_workload("C2", n=700)
    X, y, theta = w["X"].astype(dtype), w["y"].astype(dtype), w["theta"].copy()
    if dtype == np.float32:
        theta[0] = theta[1] + math.log(0.5)
    s2, c, ell = split_theta(theta)
    fk = gpr.FittedKernel.extend(X, y, theta)
    Xs = synth.candidates("C2", 40, w["d"]).astype(dtype)
    ref = O.extend(X.astype(np.float64), y.astype(np.float64), s2, c, ell, 2.5)
    rm, rv, _ = O.predict(Xs.astype(np.float64), X.astype(np.float64), ref["alpha"], ref["k_inv"], c, ell, 2.5)
    tol = F64_TOL if dtype == np.float64 else F32_TOL
    big_mean, big_var, _ = fk.predict(Xs)  # 40 candidates: the batched (tile GEMM) path
    for start in (0, 7, 24):
        mean, var, n_warn = fk.predict(Xs[start:start + m])
        assert mean.dtype == dtype and n_warn == 0
        np.testing.assert_allclose(mean, rm[start:start + m], rtol=0, atol=tol * max(1.0, np.abs(rm).max()))
        np.testing.assert_allclose(var, np.maximum(rv[start:start + m], 0), rtol=0, atol=tol * c)
        np.testing.assert_allclose(mean, big_mean[start:start + m], rtol=0, atol=(1e-12 if dtype == np.float64 else 2e-5) * max(1.0, np.abs(rm).max()))
        np.testing.assert_allclose(var, big_var[start:start + m], rtol=0, atol=(1e-12 if dtype == np.float64 else 2e-5) * c)
        mean_only, none, _ = fk.predict(Xs[start:start + m], want_variance=False)
        assert none is None and np.array_equal(mean_only, mean)
    # bitwise reproducible
    a = fk.predict(Xs[:m])
    b = fk.predict(Xs[:m])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def _random_cases(count, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(count):
        n = int(rng.integers(1, 700))
        d = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 31, 64]))
        nu = [0.5, 1.5, 2.5, math.inf][int(rng.integers(0, 4))]
        dtype = np.float32 if rng.random() < 0.25 else np.float64
        cases.append((n, d, nu, dtype, int(rng.integers(0, 2 ** 31))))
    return cases


@pytest.mark.parametrize("n,d,nu,dtype,seed", _random_cases(24, 20261004))
def test_random_shapes_against_oracle(n, d, nu, dtype, seed):
    # A seeded sweep over what the fixed cases leave out: ragged n (1 .. 699, every padding remainder), every d class up to the
    # 64-feature limit, all four kernels, both element types -- lml, gradient, alpha and the predictive mean / variance at a
    # few candidates against the oracle (lml.rs:29-79, predict.rs:7-52) at the path's bar: 1e-8 (f64), 1e-4 (f32, against the
    # f64 oracle on the same rounded inputs).
    rng = np.random.default_rng(seed)
    X = rng.random((n, d)).astype(dtype)
    y = (np.sin(3.0 * X[:, 0].astype(np.float64)) + 0.3 * rng.standard_normal(n) + 1.0).astype(dtype)
    ell = 0.4 + 0.6 * rng.random(d) * math.sqrt(d)
    c = 0.5 + rng.random()
    s2 = c * (0.05 + 0.2 * rng.random())  # well-conditioned: the sweep is about shapes, not about cond(K)
    theta = np.concatenate([[math.log(s2), math.log(c)], np.log(ell)])
    tol = 1e-8 if dtype == np.float64 else 1e-4
    X64, y64 = X.astype(np.float64), y.astype(np.float64)
    ref = O.lml_with_gradient(X64, y64, s2, c, ell, nu)
    prob = gpr.Problem(X, y, nu=nu)
    lml, grad = prob.lml_with_gradient(theta)
    alpha, kinv, _ = prob.results()
    prob.close()
    assert abs(lml - ref["lml"]) <= tol * max(1.0, abs(ref["lml"]))
    np.testing.assert_allclose(grad, ref["grad"], rtol=0, atol=tol * max(1.0, np.abs(ref["grad"]).max()))
    np.testing.assert_allclose(alpha, ref["alpha"], rtol=0, atol=tol * max(1.0, np.abs(ref["alpha"]).max()))
    np.testing.assert_allclose(kinv, ref["k_inv"], rtol=0, atol=tol * max(1.0, np.abs(ref["k_inv"]).max()))
    fk = gpr.FittedKernel.extend(X, y, theta, nu=nu)
    Xs = rng.random((5, d)).astype(dtype)
    mean, var, _ = fk.predict(Xs)
    rm, rv, _ = O.predict(Xs.astype(np.float64), X64, ref["alpha"], ref["k_inv"], c, ell, nu)
    fk.release()
    np.testing.assert_allclose(mean, rm, rtol=0, atol=tol * max(1.0, np.abs(rm).max()))
    np.testing.assert_allclose(var, rv, rtol=0, atol=tol * c * (10 if dtype == np.float32 else 1))


@pytest.mark.parametrize("n,dtype", [(300, np.float64), (300, np.float32), (2300, np.float64)])
def test_nan_features_stay_nan(n, dtype):
    # the fast fp64 sqrt / exp of kmat / gradtrace / kstar used to turn a NaN argument into distance 0 (K entry = c), so a
    # NaN in X or in the candidates silently gave a finite model.  The reference propagates it: the Cholesky fails and the
    # objective is +inf (lml.rs:47-50); a NaN candidate predicts NaN (predict.rs:18-37).  Both element types, both evaluation paths.
    w = synth.make_workload("C2", n=n)
    X, y, theta = w["X"].astype(dtype), w["y"].astype(dtype), w["theta"]
    bad = X.copy()
    bad[17, 1] = np.nan
    prob = gpr.Problem(bad, y)
    assert prob.lml_with_gradient(theta) is None  # NOT_PD: objective +inf, zero gradient (fit.rs:105-112)
    prob.close()
    with pytest.raises(gpr.HbegpError):
        gpr.FittedKernel.extend(bad, y, theta)  # "Kernel matrix must be invertible." (fit.rs:55)
    fk = gpr.FittedKernel.extend(X, y, theta)
    for m in (5, 64):  # the path for a handful of candidates and the batched path
        Xs = synth.candidates("C2", m, X.shape[1]).astype(dtype)
        Xs[3, 0] = np.nan
        mean, var, _ = fk.predict(Xs)
        assert np.isnan(mean[3]) and np.isnan(var[3])
        ok = np.arange(m) != 3
        assert np.all(np.isfinite(mean[ok])) and np.all(np.isfinite(var[ok]))
    fk.release()


@pytest.mark.parametrize("n,d,nu,dtype", [(128, 8, 2.5, np.float64), (100, 2, 1.5, np.float64), (64, 32, 0.5, np.float64), (17, 3, math.inf, np.float64),
                                           (1, 1, 2.5, np.float64), (128, 5, 2.5, np.float32), (77, 16, 1.5, np.float32)])
def test_single_launch_path_for_the_reference_regime(n, d, nu, dtype, monkeypatch):
    # Up to 128 rows (the reference's own regime, minimize.rs:118-120) one evaluation is ONE launch that keeps K, L, L^-1, K^-1 and
    # alpha in the LDS (small_eval_kernel).  It must agree with the general five-launch path (HBEGP_SMALL=0) far inside the bar
    # -- same kernel matrix and factor, other summation orders behind them -- and with the oracle at the bar; fit, extend and
    # predict run on top of it.
    rng = np.random.default_rng(1000 * n + d)
    X = rng.random((n, d)).astype(dtype)
    y = (np.sin(3.0 * X[:, 0].astype(np.float64)) + 0.3 * rng.standard_normal(n) + 1.0).astype(dtype)
    ell = 0.4 + 0.6 * rng.random(d) * math.sqrt(d)
    c = 0.5 + rng.random()
    s2 = c * (0.05 + 0.2 * rng.random())
    theta = np.concatenate([[math.log(s2), math.log(c)], np.log(ell)])
    tol = 1e-8 if dtype == np.float64 else 1e-4

    def run():
        prob = gpr.Problem(X, y, nu=nu)
        lml, grad = prob.lml_with_gradient(theta)
        alpha, kinv, ldiag = prob.results()
        prob.close()
        fk = gpr.FittedKernel.extend(X, y, theta, nu=nu)
        Xs = rng.random((5, d)).astype(dtype)
        mean, var, _ = fk.predict(Xs)
        a2, k2 = fk.arrays()
        fk.release()
        return dict(lml=lml, grad=grad, alpha=alpha, kinv=kinv, ldiag=ldiag, mean=mean, var=var, Xs=Xs, a2=a2, k2=k2)

    rng_state = rng.bit_generator.state
    small = run()
    rng.bit_generator.state = rng_state
    monkeypatch.setenv("HBEGP_SMALL", "0")
    general = run()
    cmp_tol = 1e-12 if dtype == np.float64 else 2e-5
    for key in ("lml", "grad", "alpha", "kinv", "ldiag", "mean", "var", "a2", "k2"):
        a, b = np.asarray(small[key], dtype=np.float64), np.asarray(general[key], dtype=np.float64)
        assert np.all(np.isfinite(a))
        assert float(np.max(np.abs(a - b))) <= cmp_tol * max(1.0, float(np.max(np.abs(b)))), key
    X64, y64 = X.astype(np.float64), y.astype(np.float64)
    ref = O.lml_with_gradient(X64, y64, s2, c, ell, nu)
    assert abs(small["lml"] - ref["lml"]) <= tol * max(1.0, abs(ref["lml"]))
    np.testing.assert_allclose(small["grad"], ref["grad"], rtol=0, atol=tol * max(1.0, np.abs(ref["grad"]).max()))
    np.testing.assert_allclose(small["alpha"], ref["alpha"], rtol=0, atol=tol * max(1.0, np.abs(ref["alpha"]).max()))
    np.testing.assert_allclose(small["kinv"], ref["k_inv"], rtol=0, atol=tol * max(1.0, np.abs(ref["k_inv"]).max()))
    rm, rv, _ = O.predict(small["Xs"].astype(np.float64), X64, ref["alpha"], ref["k_inv"], c, ell, nu)
    np.testing.assert_allclose(small["mean"], rm, rtol=0, atol=tol * max(1.0, np.abs(rm).max()))
    np.testing.assert_allclose(small["var"], rv, rtol=0, atol=tol * c * (10 if dtype == np.float32 else 1))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fresh_pool_blocks_are_cleared_before_their_first_writer(dtype, monkeypatch):
    # Round 4's intermittent wrong variance (profiles/r04_memset_race.txt), made deterministic.  A fresh pool block is cleared
    # once; hipMemset on device memory is queued on the NULL stream and returns at once, while the engine's streams are
    # non-blocking ones that do not wait for the null stream -- the clear could land on top of (or concurrently with) the block's
    # first writer.  HBEGP_POOL_FRESH=1 makes every block a fresh one (as in a fresh process), HBEGP_POOL_NULL_DELAY_MB queues a
    # long fill in front of every clear so that an unsynchronised clear is LATE for certain (tests/pool_clear_scenario.py).
    #   (a) the shipped clear (a stream of the pool's own, waited for): oracle results at the plain bar, in this process;
    #   (b) HBEGP_POOL_OLD_CLEAR=1 (round 1-4's hipMemset alone), in a FRESH process: the model is damaged -- if it were NOT, the
    #       diagnosis of round 4 would be wrong and that failure still open.  A fresh process because the damage needs the
    #       model's stream on another hardware queue than the null stream's (on the same queue it runs behind the clear and is
    #       safe by accident); HIP deals queues in stream-creation order, which only a fresh process fixes.  Inside the suite's
    #       process this leg failed to damage in about one run of six, and under eager launches always.
    import json
    import subprocess
    import sys

    import pool_clear_scenario as S

    tol = F64_TOL if dtype == np.float64 else F32_TOL
    monkeypatch.setenv("HBEGP_POOL_FRESH", "1")
    monkeypatch.setenv("HBEGP_POOL_NULL_DELAY_MB", "8192")
    run = S.scenario(dtype)
    devs = run()
    assert max(devs) <= tol, devs
    env = dict(os.environ, HBEGP_POOL_OLD_CLEAR="1")
    for k in ("HBEGP_NO_GRAPH", "GPU_MAX_HW_QUEUES"):  # (tools/gpu_matrix.sh: the demonstration is about the shipped configuration)
        env.pop(k, None)
    cp = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "pool_clear_scenario.py"), np.dtype(dtype).name],
                        env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
    assert lines, (cp.stdout[-500:], cp.stderr[-500:])
    old = json.loads(lines[-1])
    damaged = "error" in old or not (max(old["deviations"]) <= tol)  # NaN counts as damaged
    again = run()
    print(f"old clear ({np.dtype(dtype).name}, fresh process): {old}; shipped clear: {devs}, again {again}")
    assert damaged, f"the unsynchronised clear did not damage the model ({old}): round 4's diagnosis does not hold"
    assert max(again) <= tol, again
