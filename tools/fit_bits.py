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
# FIT_BITS_MIX=1: other fits in between (other sizes, f32) so that the pooled work matrices change roles and carry other data
mix = os.environ.get("FIT_BITS_MIX")
import numpy as np
rng = np.random.default_rng(3)
for it in range(nfit):
    if mix and it:
        m = int(rng.integers(900, 4096))
        w2 = synth.make_workload("M", n=m)
        dt = np.float32 if it % 3 == 0 else np.float64
        fk2 = gpr.FittedKernel.new(w2["X"].astype(dt), w2["y"].astype(dt), w2["theta0"], w2["lo"], w2["hi"], synth.restart_points("M", w2["lo"], w2["hi"], 2), nu=2.5, ctx=ctx, maxeval=7)
        fk2.release()
    fk = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], starts, nu=2.5, ctx=ctx, maxeval=150, fixed_work=True)
    vals.append(fk.lml)
    fk.release()
print(len(set(vals)), "distinct values in", nfit, "fits:", sorted(set(vals)))
