#!/usr/bin/env python3
"""Soak test: many fits / extends / predicts of changing shapes in one process; checks results stay finite, the model
reproduces its training data roughly, and device memory does not creep (pool bounded, no leaks)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device memory query only)

from hbetune_rs_amd import gpr, synth  # noqa: E402

rng = np.random.default_rng(1)
free0 = None
t0 = time.time()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for it in range(iters):
    n = int(rng.integers(3, 1800))
    d = int(rng.integers(1, 12))
    dtype = np.float32 if it % 7 == 3 else np.float64
    X = rng.random((n, d)).astype(dtype)
    y = (np.sin(3 * X).sum(axis=1) + 0.05 * rng.standard_normal(n)).astype(dtype)
    y = (y - y.min()) / max(1e-6, y.mean() - y.min()) + 0.05
    p = d + 2
    lo = np.array([1e-5, 1e-2] + [1e-3] * d)
    hi = np.array([1e5, 1e2] + [1e3] * d)
    theta0 = np.log(np.array([1e-2, 1.0] + [0.5] * d))
    starts = np.log(lo) + rng.random((2, p)) * (np.log(hi) - np.log(lo))
    fk = gpr.FittedKernel.new(X, y, theta0, lo, hi, starts=starts, maxeval=12, nu=[0.5, 1.5, 2.5][it % 3])
    mean, var, _ = fk.predict(X[: min(n, 200)])
    assert np.isfinite(mean).all() and np.isfinite(var).all() and (var >= 0).all(), (it, n, d)
    fk2 = gpr.FittedKernel.extend(X, y, fk.theta, lo, hi, nu=[0.5, 1.5, 2.5][it % 3])
    m2, _, _ = fk2.predict(X[: min(n, 200)], want_variance=False)
    tol = 1e-2 if dtype == np.float32 else 1e-6
    assert np.allclose(m2, mean, rtol=tol, atol=tol), (it, n, d, np.abs(m2 - mean).max())
    del fk, fk2
    free, total = torch.cuda.mem_get_info()
    if it == 10:
        free0 = free
    if it % 10 == 0:
        print(f"iter {it}: n={n} d={d} {np.dtype(dtype).name} free={free / 2**30:.2f} GiB  t={time.time() - t0:.1f}s", flush=True)
free, _ = torch.cuda.mem_get_info()
print(f"done: free after warm-up {free0 / 2**30:.2f} GiB, at end {free / 2**30:.2f} GiB")
# the pool may hold up to its cap; what must not happen is unbounded growth
assert free0 - free < 26 * 2**30, "device memory keeps growing"
print("SOAK OK")
