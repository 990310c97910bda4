#!/usr/bin/env python3
"""Run-to-run reproducibility of the fixed-work fit of config M: prints the fitted lml of each of N fits (all must agree)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth
nfit = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else None
w = synth.make_workload("M", n=n)
starts = synth.restart_points("M", w["lo"], w["hi"], 2)
ctx = gpr.Context(device_ids=[0])
vals = []
for _ in range(nfit):
    fk = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], starts, nu=2.5, ctx=ctx, maxeval=150, fixed_work=True)
    vals.append(fk.lml)
    fk.release()
print(len(set(vals)), "distinct values in", nfit, "fits:", sorted(set(vals)))
