#!/usr/bin/env python3
"""Write a task trace of one evaluation (HBEGP_DAG_TRACE must name the output file) and analyse it.
Usage: HBEGP_DAG_TRACE=out.txt [HBEGP_DAG_SPLIT=1 ...] trace_eval.py [n]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth
os.environ.setdefault("HBEGP_DAG", "1")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = synth.make_workload("M", n=n)
prob = gpr.Problem(w["X"], w["y"])
ph = prob.time_eval(w["theta"], reps=3)
print({k: round(v, 3) for k, v in ph.items() if k in ("eval_graph_ms", "dag_ms", "lauum_ms", "kmat_ms", "alpha_ms", "gradtrace_ms")})
prob.close()
subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "dag_trace.py"), os.environ["HBEGP_DAG_TRACE"]])
