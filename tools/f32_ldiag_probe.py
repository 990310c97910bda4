#!/usr/bin/env python3
"""Signed bias of diag(L) (device f32 vs LAPACK f32 vs f64 truth), per 128-block."""
import math, os, sys
import numpy as np
import scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth
from oracle import gpr_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w = synth.make_workload("C5", n=n)
theta = w["theta"].copy()
s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
X64 = w["X"].astype(np.float64)
K64 = O.product_kernel(X64, X64, c, ell, 2.5) + s2 * np.eye(n)
K32 = (O.product_kernel(w["X"], w["X"], np.float32(c), ell, 2.5) + np.float32(s2) * np.eye(n, dtype=np.float32)).astype(np.float32)
L64 = np.linalg.cholesky(K64)
L32of32 = sl.cholesky(K32, lower=True)                      # LAPACK spotrf on the f32 kernel
L64of32 = np.linalg.cholesky(K32.astype(np.float64))        # exact factor of the f32 kernel
prob = gpr.Problem(w["X"], w["y"])
prob.lml_with_gradient(theta)
_, _, ld = prob.results(want_kinv=False)
def show(name, d, ref):
    r = d.astype(np.float64) / ref - 1
    print(f"{name:28s} mean {r.mean():+.3e} rms {np.sqrt((r*r).mean()):.3e}  per block:", " ".join("%+.1e" % r[b:b+128].mean() for b in range(0, n, 128)))
show("device vs f64 truth", ld, np.diag(L64))
show("device vs exact(f32 kernel)", ld, np.diag(L64of32))
show("lapack32 vs f64 truth", np.diag(L32of32), np.diag(L64))
show("lapack32 vs exact(f32 kernel)", np.diag(L32of32), np.diag(L64of32))
show("exact(f32 kernel) vs truth", np.diag(L64of32), np.diag(L64))
