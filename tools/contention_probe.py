#!/usr/bin/env python3
"""Does a task-queue launch slow down when two more run beside it?  (is the fit's 0.72 of its share
per-task overhead or fabric / HBM contention?)

  A. one slot alone, its launch sized like a fit's (96 workgroups = the 3-busy-slot variant), quiet device
  B. three slots at once, 96 workgroups each (hbegp_problem_time_concurrent: what bench.py's roofline uses)
  C. one slot alone with all 256 workgroups (chain-bound)
Same task set, same queue order in A and B; the launch's duration comes from hipEvents on the slot's stream (eager pass).
usage: contention_probe.py [n=4096]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = synth.make_workload("M", n=n)
os.environ["HBEGP_DAG"] = "1"
os.environ["HBEGP_DAG_LAUUM_SPLIT"] = "0"  # same task set in A, B and C
out = {"n": n}
p3 = gpr.Problem(w["X"], w["y"], n_slots=3)
b = p3.time_concurrent(w["theta"], reps=8)
p3.close()
wg = int(b["task_queue_workgroups"])
out["B_three_slots"] = {"workgroups_each": wg, "dag_ms": b["factor_ms"], "round_ms_graph": b["round_ms"], "round_ms_eager": b["round_eager_ms"],
                        "kmat_ms": b["kmat_ms"], "alpha_ms": b["alpha_ms"], "gradtrace_ms": b["gradtrace_ms"], "gflop": b["factor_gflop"]}
os.environ["HBEGP_DAG_WG"] = str(wg)
p1 = gpr.Problem(w["X"], w["y"], n_slots=1)
a = p1.time_eval(w["theta"], reps=8)
p1.close()
out["A_one_slot_same_size"] = {"workgroups": wg, "dag_ms": a["dag_ms"], "eval_graph_ms": a["eval_graph_ms"], "kmat_ms": a["kmat_ms"],
                               "alpha_ms": a["alpha_ms"], "gradtrace_ms": a["gradtrace_ms"]}
del os.environ["HBEGP_DAG_WG"]
p1 = gpr.Problem(w["X"], w["y"], n_slots=1)
c = p1.time_eval(w["theta"], reps=8)
p1.close()
out["C_one_slot_whole_chip"] = {"workgroups": 256, "dag_ms": c["dag_ms"], "eval_graph_ms": c["eval_graph_ms"]}
out["slowdown_B_over_A"] = b["factor_ms"] / a["dag_ms"] if a["dag_ms"] else None
gf = b["factor_gflop"]
out["tflops"] = {"A": gf / a["dag_ms"], "B_each": gf / b["factor_ms"], "B_aggregate": 3 * gf / b["factor_ms"], "C": gf / c["dag_ms"]}
print(json.dumps(out, indent=1))
