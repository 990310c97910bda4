#!/usr/bin/env python3
"""f32 path: deviation of the GPU results and of the f32 LAPACK oracle from the f64 oracle (same X, y, theta)."""
import json, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth  # noqa: E402
from oracle import gpr_oracle as O  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for jit in (0.5, 0.05, 0.01):
    w = synth.make_workload("C5", n=n)
    theta = w["theta"].copy(); theta[0] = theta[1] + math.log(jit)
    s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
    X64, y64 = w["X"].astype(np.float64), w["y"].astype(np.float64)
    r64 = O.lml_with_gradient(X64, y64, s2, c, ell, 2.5)
    r32 = O.lml_with_gradient(w["X"], w["y"], s2, c, ell, 2.5)
    prob = gpr.Problem(w["X"], w["y"])
    lml, grad = prob.lml_with_gradient(theta)
    alpha, kinv, _ = prob.results()
    fk = gpr.FittedKernel.extend(w["X"], w["y"], theta)
    Xs = synth.candidates("C5", 128, 2).astype(np.float32)
    mean, var, _ = fk.predict(Xs)
    m64, v64, _ = O.predict(Xs.astype(np.float64), X64, r64["alpha"], r64["k_inv"], c, ell, 2.5)
    m32, v32, _ = O.predict(Xs, w["X"], r32["alpha"], r32["k_inv"], c, ell, 2.5)
    out = dict(n=n, jitter=jit, cond=float(np.linalg.cond(r64["kernel_matrix"])),
               lml_rel=(abs(lml - r64["lml"]) / abs(r64["lml"]), abs(r32["lml"] - r64["lml"]) / abs(r64["lml"])),
               grad_rel=(float(np.abs(grad - r64["grad"]).max() / np.abs(r64["grad"]).max()), float(np.abs(r32["grad"] - r64["grad"]).max() / np.abs(r64["grad"]).max())),
               alpha_rel=(float(np.abs(alpha - r64["alpha"]).max() / np.abs(r64["alpha"]).max()), float(np.abs(r32["alpha"] - r64["alpha"]).max() / np.abs(r64["alpha"]).max())),
               kinv_rel=(float(np.abs(kinv - r64["k_inv"]).max() / np.abs(r64["k_inv"]).max()), float(np.abs(r32["k_inv"] - r64["k_inv"]).max() / np.abs(r64["k_inv"]).max())),
               mean_rel=(float(np.abs(mean - m64).max() / np.abs(m64).max()), float(np.abs(m32 - m64).max() / np.abs(m64).max())),
               var_over_c=(float(np.abs(var - v64).max() / c), float(np.abs(v32 - v64).max() / c)))
    print(json.dumps({k: (v if not isinstance(v, tuple) else ["gpu %.2e" % v[0], "lapack32 %.2e" % v[1]]) for k, v in out.items()}))
