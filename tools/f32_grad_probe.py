#!/usr/bin/env python3
"""Where does the f32 gradient error come from?  Recompute the gradient on the host from the DEVICE's f32 alpha / K^-1
(a) in f64 arithmetic, (b) in f32 arithmetic, and compare with the device gradient and the f64 truth."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth
from oracle import gpr_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w = synth.make_workload("C5", n=n)
theta = w["theta"].copy()
s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
X64, y64 = w["X"].astype(np.float64), w["y"].astype(np.float64)
r64 = O.lml_with_gradient(X64, y64, s2, c, ell, 2.5)
r32 = O.lml_with_gradient(w["X"], w["y"], s2, c, ell, 2.5)
prob = gpr.Problem(w["X"], w["y"])
lml, grad = prob.lml_with_gradient(theta)
alpha, kinv, _ = prob.results()
K, dK = O.product_theta_grad(X64, c, ell, 2.5)  # f64 kernel + gradient tensor
def grad_from(alpha, kinv, dt):
    a = alpha.astype(dt); ki = kinv.astype(dt)
    tmp = np.outer(a, a) - ki
    g = [0.5 * float((tmp * (np.eye(n, dtype=dt) * dt(s2))).sum())]
    for j in range(dK.shape[2]):
        g.append(0.5 * float((tmp * dK[:, :, j].astype(dt)).sum()))
    return np.array(g)
scale = np.abs(r64["grad"]).max()
print("truth grad", r64["grad"])
print("device grad err/scale      ", (grad - r64["grad"]) / scale)
print("lapack32 grad err/scale    ", (r32["grad"] - r64["grad"]) / scale)
print("host f64 arith on device alpha,Kinv:", (grad_from(alpha, kinv, np.float64) - r64["grad"]) / scale)
print("host f64 arith on lapack32 alpha,Kinv:", (grad_from(r32["alpha"], r32["k_inv"], np.float64) - r64["grad"]) / scale)
# split: device alpha with exact Kinv, exact alpha with device Kinv
print("device alpha + exact Kinv:", (grad_from(alpha, r64["k_inv"], np.float64) - r64["grad"]) / scale)
print("exact alpha + device Kinv:", (grad_from(r64["alpha"], kinv, np.float64) - r64["grad"]) / scale)
print("lapack alpha + exact Kinv:", (grad_from(r32["alpha"], r64["k_inv"], np.float64) - r64["grad"]) / scale)
print("exact alpha + lapack Kinv:", (grad_from(r64["alpha"], r32["k_inv"], np.float64) - r64["grad"]) / scale)
# residual structure: diag of Kinv*K - I per 128-block (device vs lapack32), and of X-based pieces
Kfull = r64["kernel_matrix"]
for name, ki in (("device", kinv.astype(np.float64)), ("lapack32", r32["k_inv"].astype(np.float64))):
    R = ki @ Kfull - np.eye(n)
    dg = np.diag(R)
    print(name, "trace(R)/n = %.3e" % (dg.sum() / n), " per 128-block mean diag(R):", " ".join("%.1e" % dg[b:b + 128].mean() for b in range(0, n, 128)))
    print(name, " |R|_max = %.2e" % np.abs(R).max())
if os.environ.get("HBEGP_F32_REFINE", "1") != "0":
    pass
