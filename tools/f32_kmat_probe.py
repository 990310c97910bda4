#!/usr/bin/env python3
"""Bias of the f32 kernel matrix (device vs numpy-f32 oracle) against the f64 kernel of the same f32 inputs."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth
from oracle import gpr_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w = synth.make_workload("C5", n=n)
theta = w["theta"].copy()
s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
X64 = w["X"].astype(np.float64)
K64 = O.product_kernel(X64, X64, c, ell, 2.5)
K32 = O.product_kernel(w["X"], w["X"], np.float32(c), ell, 2.5).astype(np.float64)
prob = gpr.Problem(w["X"], w["y"])
Kd = prob.kernel_matrix(theta).astype(np.float64)
Kd[np.diag_indices(n)] -= np.float32(s2)
off = ~np.eye(n, dtype=bool)
for name, K in (("device", Kd), ("numpy32", K32)):
    rel = (K[off] - K64[off]) / K64[off]
    print(f"{name}: off-diagonal relative error mean {rel.mean():+.3e}  rms {np.sqrt((rel**2).mean()):.3e}  max {np.abs(rel).max():.3e};"
          f" diag err {np.abs(np.diag(K) - np.diag(K64)).max() / c:.2e}")
    # effect on the smallest eigenvalues
    ev = np.linalg.eigvalsh(K + s2 * np.eye(n))[:3]
    print("   smallest eigenvalues of K + s2 I:", ev, " truth:", np.linalg.eigvalsh(K64 + s2 * np.eye(n))[:3])
