#!/usr/bin/env python3
"""Where one bench step (fixed-work fit + predict, config M) spends its wall time: HBEGP_TIMING=1 prints the fit's phases,
this script adds the predict and the release.  usage: step_timing.py [steps]"""
import os, sys, time
os.environ["HBEGP_TIMING"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from hbetune_rs_amd import gpr, synth
w = synth.make_workload("M")
st = synth.restart_points("M", w["lo"], w["hi"], 2)
Xs = synth.candidates("M", 1600, w["X"].shape[1])
ctx = gpr.Context(device_ids=[0])
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    t0 = time.perf_counter()
    fk = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], st, nu=2.5, ctx=ctx, maxeval=150, fixed_work=True)
    t1 = time.perf_counter()
    mean, var, _ = fk.predict(Xs)
    t2 = time.perf_counter()
    fk.release()
    t3 = time.perf_counter()
    print(f"step {i}: fit {1e3 * (t1 - t0):.2f} ms, predict {1e3 * (t2 - t1):.2f} ms, release {1e3 * (t3 - t2):.2f} ms, total {1e3 * (t3 - t0):.2f} ms", flush=True)
ctx.close()
