"""Repeat one f32 (or f64) extend + small-batch predict many times and compare every result with the first, bit for bit.
usage: flaky_probe.py [reps] [n] [d] [dtype]   (HBEGP_LIB selects the library build)"""
import math
import sys

import numpy as np

sys.path.insert(0, ".")
from hbetune_rs_amd import gpr  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 589
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dtype = np.float32 if (len(sys.argv) <= 4 or sys.argv[4] == "f32") else np.float64
rng = np.random.default_rng(17349209)
X = rng.random((n, d)).astype(dtype)
y = (np.sin(3.0 * X[:, 0].astype(np.float64)) + 0.3 * rng.standard_normal(n) + 1.0).astype(dtype)
ell = 0.4 + 0.6 * rng.random(d) * math.sqrt(d)
c = 0.5 + rng.random()
s2 = c * (0.05 + 0.2 * rng.random())
theta = np.concatenate([[math.log(s2), math.log(c)], np.log(ell)])
Xs = rng.random((5, d)).astype(dtype)
first = None
bad = 0
for r in range(reps):
    fk = gpr.FittedKernel.extend(X, y, theta, nu=1.5)
    mean, var, _ = fk.predict(Xs)
    alpha, kinv = fk.arrays()
    fk.release()
    cur = (mean.copy(), var.copy(), alpha.copy(), kinv.copy())
    if first is None:
        first = cur
        print("first var", var)
        continue
    diffs = [not np.array_equal(a, b) for a, b in zip(cur, first)]
    if any(diffs):
        bad += 1
        print(f"rep {r}: differs in mean/var/alpha/kinv = {diffs}; var {var}; max|dalpha| {np.abs(cur[2]-first[2]).max():.3e} max|dkinv| {np.abs(cur[3]-first[3]).max():.3e}")
print(f"{bad} deviating repetitions of {reps}")
