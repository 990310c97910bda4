#!/usr/bin/env python3
"""C3 (rastrigin d=16, n=4096, 8 optimiser runs) on one GPU: wall time per fixed-work fit for different numbers of
concurrently running optimiser runs (HBEGP_MAX_CONCURRENT)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth  # noqa: E402
maxeval = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w = synth.make_workload("C3")
starts = synth.restart_points("C3", w["lo"], w["hi"], 7)
fk = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], starts, maxeval=2, fixed_work=True); fk.release()
t0 = time.perf_counter()
fk = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], starts, maxeval=maxeval, fixed_work=True)
dt = time.perf_counter() - t0
print(json.dumps({"conc": os.environ.get("HBEGP_MAX_CONCURRENT", "4"), "evals": 8 * maxeval, "fit_s": dt, "ms_per_eval": dt / (8 * maxeval) * 1e3, "lml": fk.lml}))
