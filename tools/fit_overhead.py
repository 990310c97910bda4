#!/usr/bin/env python3
"""Fixed cost of a fit call (workspace allocation, graph capture, model construction): maxeval=1 vs maxeval=150."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth  # noqa: E402
w = synth.make_workload("M")
starts = synth.restart_points("M", w["lo"], w["hi"], 2)
out = {}
for me in (1, 1, 1, 20):
    t0 = time.perf_counter()
    fk = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], starts, maxeval=me, fixed_work=True)
    out.setdefault(f"fit_maxeval{me}_ms", []).append(round((time.perf_counter() - t0) * 1e3, 2))
    fk.release()
print(json.dumps(out))
