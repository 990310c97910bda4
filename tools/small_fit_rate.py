#!/usr/bin/env python3
"""Fixed-work fits per second at small n (the reference's regime): 3 optimiser runs x 150 evaluations, config M data cut to n rows.
usage: small_fit_rate.py [n ...]   (HBEGP_SMALL=0 for the general five-launch path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth
for n in [int(a) for a in sys.argv[1:]] or [64, 100, 128, 200, 256]:
    w = synth.make_workload("M", n=n)
    st = synth.restart_points("M", w["lo"], w["hi"], 2)
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        f = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], st, nu=2.5, maxeval=150, fixed_work=True)
        dt = time.perf_counter() - t0
        f.release()
        best = dt if best is None else min(best, dt)
    prob = gpr.Problem(w["X"], w["y"])
    ph = prob.time_eval(w["theta"], reps=50)
    prob.close()
    print(f"n={n}: {1.0 / best:.1f} fixed-work fits/s ({450 / best:.0f} evaluations/s); one evaluation back to back {ph['eval_graph_ms'] * 1e3:.1f} us")
