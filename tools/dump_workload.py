#!/usr/bin/env python3
"""Writes config M's data cut to n rows (features, targets, start point, box, restart points) as one binary file for
tools/concurrent_fits_native.cpp: int32 n, d, n_restarts, pad; then float64 X[n*d], y[n], theta0[p], lo[p], hi[p], starts[n_restarts*p]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import synth  # noqa: E402

n, path = int(sys.argv[1]), sys.argv[2]
w = synth.make_workload("M", n=n)
starts = synth.restart_points("M", w["lo"], w["hi"], 2)
with open(path, "wb") as f:
    f.write(np.array([n, w["d"], starts.shape[0], 0], dtype=np.int32).tobytes())
    for a in (w["X"], w["y"], w["theta0"], w["lo"], w["hi"], starts):
        f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
