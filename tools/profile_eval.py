#!/usr/bin/env python3
"""Per-phase timing of one lml+gradient evaluation (hipEvents, eager) + graph replay time.  Usage: profile_eval.py [cfg] [n]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "M"
n = int(sys.argv[2]) if len(sys.argv) > 2 else None
w = synth.make_workload(cfg, n=n)
prob = gpr.Problem(w["X"], w["y"])
ph = prob.time_eval(w["theta"], reps=5)
env = {k: v for k, v in os.environ.items() if k.startswith("HBEGP_")}
print(json.dumps({"cfg": cfg, "n": w["n"], "env": env, **{k: round(v, 4) for k, v in ph.items()}}))
