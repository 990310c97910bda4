#!/usr/bin/env python3
"""Quick rate probe for configuration sweeps: fixed-work fits of config M (no predict, no roofline extras).
Usage: fit_rate.py [fits [n]]   -> prints fits/s, ms per fit, amortised ms per evaluation; env switches apply."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth

nfit = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else None
w = synth.make_workload("M", n=n)
X, y = w["X"], w["y"]
starts = synth.restart_points("M", w["lo"], w["hi"], 2)
ctx = gpr.Context(device_ids=[0])
def fit():
    fk = gpr.FittedKernel.new(X, y, w["theta0"], w["lo"], w["hi"], starts, nu=2.5, ctx=ctx, maxeval=150, fixed_work=True)
    ne = fk.n_evals
    lml = fk.lml
    fk.release()
    return ne, lml
fit()
t0 = time.perf_counter()
for _ in range(nfit):
    ne, lml = fit()
dt = (time.perf_counter() - t0) / nfit
print(f"{1.0 / dt:.4f} fits/s  {dt * 1e3:.1f} ms/fit  {dt * 1e3 / ne:.4f} ms/eval  lml={lml!r}", flush=True)
ctx.close()
