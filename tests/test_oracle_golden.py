"""The numpy/LAPACK oracle against (a) the reference's own known-answer tables and (b) sklearn goldens."""
import glob
import json
import math
import os

import numpy as np
import pytest

from oracle import gpr_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))


@pytest.mark.parametrize("name", ["matern_nu15", "matern_nu25", "product_const_matern25"])
def test_reference_kernel_kats(name):
    kat = KATS[name]
    x = np.array(kat["x"])
    k, g = O.product_theta_grad(x, kat["amplitude"], kat["length_scale"], kat["nu"])
    want_g = np.array(kat["gradient"])
    if want_g.shape[2] == g.shape[2] - 1:  # pure-Matern tables have no amplitude slice
        g = g[:, :, 1:]
    np.testing.assert_allclose(k, np.array(kat["kernel"]), atol=kat["tol"])
    np.testing.assert_allclose(g, want_g, atol=kat["tol"])
    np.testing.assert_allclose(O.product_diag(x, kat["amplitude"]), np.diag(k), atol=1e-3)
    # the tables carry 8 digits: the restatement matches all of them
    np.testing.assert_allclose(k, np.array(kat["kernel"]), rtol=2e-7, atol=5e-9)


def test_reference_cdist_kat():
    for case in KATS["cdist"]:
        got = O.cdist(np.array(case["a"]), np.array(case["b"]))
        assert np.array_equal(got, np.sqrt(np.array(case["sq"])))


def test_reference_clamp_kat():
    for case in KATS["clamp"]:
        v = np.array(case["variances"])
        below = O.clamp_negative_variance(v, case["level"])
        assert below == case["below"]
        assert np.array_equal(v, np.array(case["after"]))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_matches_sklearn_golden(path):
    g = np.load(path)
    X, y, theta, nu = g["X"], g["y"], g["theta"], float(g["nu"])
    s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
    res = O.lml_with_gradient(X, y, s2, c, ell, nu)
    assert res is not None
    scale = max(1.0, abs(float(g["lml"])))
    assert abs(res["lml"] - float(g["lml"])) <= 1e-10 * scale
    np.testing.assert_allclose(res["grad"], g["grad"], rtol=0, atol=1e-9 * max(1.0, np.abs(g["grad"]).max()))
    np.testing.assert_allclose(res["alpha"], g["alpha"], rtol=0, atol=1e-9 * np.abs(g["alpha"]).max())
    np.testing.assert_allclose(res["kernel_matrix"][:4], g["K_rows"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(np.diag(res["kernel_matrix"]), g["K_diag"], rtol=1e-13)
    mean, var, _ = O.predict(g["Xs"], X, res["alpha"], res["k_inv"], c, ell, nu)
    np.testing.assert_allclose(mean, g["mean"], rtol=0, atol=1e-9 * max(1.0, np.abs(g["mean"]).max()))
    np.testing.assert_allclose(var, g["var"], rtol=0, atol=1e-9 * c)


def test_objective_contract_not_pd():
    # fit.rs:105-112: failed factorisation -> +inf and zero gradient.  Duplicate rows with ~zero noise.
    X = np.array([[0.1, 0.2], [0.1, 0.2], [0.5, 0.5]])
    y = np.array([1.0, 2.0, 3.0])
    theta = np.array([math.log(1e-300), 0.0, 0.0, 0.0])
    f, grad, res = O.objective(theta, X, y, 2.5, [(0, math.inf), (1e-3, 1e3), (1e-3, 1e3), (1e-3, 1e3)])
    assert f == math.inf and not grad.any() and res is None


def test_f32_oracle_tracks_f64():
    g = np.load(os.path.join(GOLDEN, "c5_himmelblau_n128.npz"))
    theta = g["theta"]
    s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
    r32 = O.lml_with_gradient(g["X"].astype(np.float32), g["y"].astype(np.float32), s2, c, ell, 2.5)
    assert r32["alpha"].dtype == np.float32
    assert abs(r32["lml"] - float(g["lml"])) < 5e-2


# ---- the plain-C oracle (no LAPACK) against the same goldens and against the numpy/LAPACK oracle -------------------
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_c_oracle_matches_sklearn_golden(path):
    from oracle import c_oracle

    g = np.load(path)
    X, y, theta, nu = g["X"], g["y"], g["theta"], float(g["nu"])
    s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
    res = c_oracle.lml_with_gradient(X, y, s2, c, ell, nu)
    assert res is not None
    assert abs(res["lml"] - float(g["lml"])) <= 1e-10 * max(1.0, abs(float(g["lml"])))
    np.testing.assert_allclose(res["grad"], g["grad"], rtol=0, atol=1e-9 * max(1.0, np.abs(g["grad"]).max()))
    np.testing.assert_allclose(res["alpha"], g["alpha"], rtol=0, atol=1e-9 * np.abs(g["alpha"]).max())
    mean, var, warn = c_oracle.predict(g["Xs"], X, res["alpha"], res["k_inv"], c, ell, nu)
    np.testing.assert_allclose(mean, g["mean"], rtol=0, atol=1e-9 * max(1.0, np.abs(g["mean"]).max()))
    np.testing.assert_allclose(var, g["var"], rtol=0, atol=1e-9 * c)
    assert warn == 0
    ref = O.lml_with_gradient(X, y, s2, c, ell, nu)
    np.testing.assert_allclose(res["k_inv"], ref["k_inv"], rtol=0, atol=1e-9 * np.abs(ref["k_inv"]).max())


def test_c_oracle_not_pd():
    from oracle import c_oracle

    X = np.array([[0.1, 0.2], [0.1, 0.2], [0.5, 0.5]])
    y = np.array([1.0, 2.0, 3.0])
    assert c_oracle.lml_with_gradient(X, y, 1e-300, 1.0, [1.0, 1.0], 2.5) is None
