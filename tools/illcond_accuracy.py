#!/usr/bin/env python3
"""Accuracy on ill-conditioned kernel matrices (small noise): GPU engine and LAPACK (the oracle) against an extended-
precision (80-bit long double) Cholesky solve.  The engine's TRSM multiplies by explicit inverses of diagonal parts, so
this is where it could lose digits relative to LAPACK."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth  # noqa: E402
from oracle import gpr_oracle as O  # noqa: E402


def chol_ld(K):
    n = K.shape[0]
    L = K.astype(np.longdouble).copy()
    for k in range(n):
        L[k, k] = np.sqrt(L[k, k])
        L[k + 1:, k] /= L[k, k]
        L[k + 1:, k + 1:] -= np.outer(L[k + 1:, k], L[k + 1:, k])
    return np.tril(L)


def solve_ld(L, b):
    n = L.shape[0]
    y = b.astype(np.longdouble).copy()
    for i in range(n):
        y[i] = (y[i] - L[i, :i] @ y[:i]) / L[i, i]
    for i in range(n - 1, -1, -1):
        y[i] = (y[i] - L[i + 1:, i] @ y[i + 1:]) / L[i, i]
    return y


n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
w = synth.make_workload("C2", n=n)
X, y = w["X"], w["y"]
for noise_rel, ell_scale in ((1e-2, 1.0), (1e-4, 4.0), (1e-6, 4.0), (1e-8, 8.0), (1e-10, 8.0)):
    theta = w["theta"].copy()
    theta[0] = theta[1] + np.log(noise_rel)
    theta[2:] += np.log(ell_scale)  # long length scales: smooth, nearly singular kernel matrix
    s2, c, ell = np.exp(theta[0]), np.exp(theta[1]), np.exp(theta[2:])
    K = O.product_kernel(X, X, c, ell, 2.5) + s2 * np.eye(n)
    cond = np.linalg.cond(K)
    L = chol_ld(K)
    alpha_x = solve_ld(L, y)
    lml_x = float(-0.5 * (y.astype(np.longdouble) @ alpha_x) - np.log(np.diag(L)).sum() - 0.5 * n * np.log(2 * np.pi))
    ref = O.lml_with_gradient(X, y, s2, c, ell, 2.5)
    prob = gpr.Problem(X, y, nu=2.5)
    lml, grad = prob.lml_with_gradient(theta)
    alpha, kinv, _ = prob.results()
    an = float(np.abs(alpha_x).max())
    out = {
        "noise/c": noise_rel, "ell_scale": ell_scale, "cond": float(cond),
        "alpha_rel_err": {"gpu": float(np.abs(alpha - alpha_x).max() / an), "lapack": float(np.abs(ref["alpha"] - alpha_x).max() / an)},
        "lml_rel_err": {"gpu": abs(lml - lml_x) / abs(lml_x), "lapack": abs(ref["lml"] - lml_x) / abs(lml_x)},
        "grad_gpu_vs_lapack_rel": float(np.abs(grad - ref["grad"]).max() / np.abs(ref["grad"]).max()),
        "kinv_gpu_vs_lapack_rel": float(np.abs(kinv - ref["k_inv"]).max() / np.abs(ref["k_inv"]).max()),
    }
    print(json.dumps(out))
