#!/usr/bin/env python3
"""Fixed-work f32 fit rates (C5 n=2048 d=2, M n=4096 d=8) and one C5 evaluation; HBEGP_LIB selects the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hbetune_rs_amd import gpr, synth
for cfg in ("C5", "M"):
    w = synth.make_workload(cfg)
    X, y = w["X"].astype(np.float32), w["y"].astype(np.float32)
    st = synth.restart_points(cfg, w["lo"], w["hi"], 2)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        f = gpr.FittedKernel.new(X, y, w["theta0"], w["lo"], w["hi"], st, nu=2.5, maxeval=150, fixed_work=True)
        dt = time.perf_counter() - t0
        f.release()
        best = dt if best is None else min(best, dt)
    prob = gpr.Problem(X, y)
    ph = prob.time_eval(w["theta"], reps=10)
    prob.close()
    print(f"{cfg} f32 n={X.shape[0]}: {1.0 / best:.3f} fixed-work fits/s; one evaluation {ph['eval_graph_ms']:.3f} ms")
