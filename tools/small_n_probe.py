#!/usr/bin/env python3
"""Small / mid n: where does one evaluation's time go?  Per n: back-to-back graph replays on one stream (device time only),
one evaluation per host round trip (what an optimiser run does), and the per-phase launch times.  usage: small_n_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hbetune_rs_amd import gpr, synth
for n in (128, 256, 512, 1024, 2048):
    w = synth.make_workload("M", n=n)
    prob = gpr.Problem(w["X"], w["y"])
    ph = prob.time_eval(w["theta"], reps=50)
    th = w["theta"].copy()
    prob.lml_with_gradient(th)
    t0 = time.perf_counter()
    reps = 300
    for r in range(reps):
        th[0] += 1e-9
        prob.lml_with_gradient(th)
    host_ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"n={n}: graph replay back to back {ph['eval_graph_ms']*1e3:.1f} us/eval; with a host round trip per evaluation {host_ms*1e3:.1f} us "
          f"(turnaround {host_ms*1e3 - ph['eval_graph_ms']*1e3:.1f} us); eager phases: kmat {ph['kmat_ms']*1e3:.1f} leaf {ph['leaf_ms']*1e3:.1f} ({ph['n_leaf']:.0f}) "
          f"gemm {ph['chol_gemm_ms']*1e3:.1f} ({ph['n_gemm']:.0f}) dag {ph['dag_ms']*1e3:.1f} lauum {ph['lauum_ms']*1e3:.1f} alpha {ph['alpha_ms']*1e3:.1f} grad {ph['gradtrace_ms']*1e3:.1f} eager total {ph['eval_eager_ms']*1e3:.1f}")
    prob.close()
