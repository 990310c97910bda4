#!/usr/bin/env python3
"""One evaluation alone: task queue (256 workgroups) vs launch path, per n.  Usage: dag_vs_launch.py n [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth
for n in [int(a) for a in sys.argv[1:]]:
    w = synth.make_workload("M", n=n)
    out = {}
    for mode, rl in (("0", "0"), ("1", "0"), ("1", "1")):
        os.environ["HBEGP_DAG"] = mode
        os.environ["HBEGP_DAG_RL"] = rl
        prob = gpr.Problem(w["X"], w["y"])
        ph = prob.time_eval(w["theta"], reps=3)
        out[mode + rl] = ph["eval_graph_ms"]
        prob.close()
    print(f"n={n}: launches {out['00']:.3f} ms, task queue: recursion {out['10']:.3f} ms, right-looking {out['11']:.3f} ms  "
          f"({n**3 * 1e-9 / min(out['10'], out['11']):.1f} TFLOP/s whole evaluation)", flush=True)
