#!/usr/bin/env python3
"""One evaluation alone through the default path (right-looking task queue) and a fixed-work fit, for the split point of the
inverse's recursion given in the environment (HBEGP_SPLIT_NUM / HBEGP_SPLIT_DEN / HBEGP_SPLIT_MIN).  usage: split_probe.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = synth.make_workload("M", n=n)
prob = gpr.Problem(w["X"], w["y"])
ph = prob.time_eval(w["theta"], reps=5)
lml, grad = prob.lml_with_gradient(w["theta"])
prob.close()
st = synth.restart_points("M", w["lo"], w["hi"], 2)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    fk = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], st, nu=2.5, maxeval=150, fixed_work=True)
    best = min(best, time.perf_counter() - t0)
    fk.release()
print(f"n={n} split {os.environ.get('HBEGP_SPLIT_NUM', '1')}/{os.environ.get('HBEGP_SPLIT_DEN', '2')} min {os.environ.get('HBEGP_SPLIT_MIN', '8')}: "
      f"one evaluation {ph['eval_graph_ms']:.3f} ms (queue launch {ph['dag_ms']:.3f}), fit {1 / best:.3f} /s, lml {lml!r}", flush=True)
