#!/usr/bin/env python3
"""Why does one evaluation per host round trip cost +146 us at n = 512 only?  usage: n512_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hbetune_rs_amd import gpr, synth
for n in (384, 448, 512, 576, 640, 768):
    w = synth.make_workload("M", n=n)
    prob = gpr.Problem(w["X"], w["y"])
    ph = prob.time_eval(w["theta"], reps=50)
    th = w["theta"].copy()
    for _ in range(20):
        prob.lml_with_gradient(th)
    ts = []
    for r in range(300):
        th[0] += 1e-9
        t0 = time.perf_counter()
        prob.lml_with_gradient(th)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"n={n} graph={os.environ.get('HBEGP_NO_GRAPH','0')=='0'}: back to back {ph['eval_graph_ms']*1e3:.1f} us; per host round trip: median {np.median(ts):.1f} min {ts.min():.1f} p90 {np.percentile(ts,90):.1f} max {ts.max():.1f}")
    prob.close()
