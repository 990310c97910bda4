#!/usr/bin/env python3
"""Fit rate and host CPU time of fixed-work fits (how much of the host a fit occupies while it waits for the GPU).
usage: cpu_cost_probe.py [n] [fits]"""
import os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = synth.make_workload("M", n=n)
st = synth.restart_points("M", w["lo"], w["hi"], 2)
ctx = gpr.Context(device_ids=[0])
def fit():
    f = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], st, nu=2.5, ctx=ctx, maxeval=150, fixed_work=True)
    f.release()
fit()
r0 = resource.getrusage(resource.RUSAGE_SELF)
t0 = time.perf_counter()
for _ in range(fits):
    fit()
dt = time.perf_counter() - t0
r1 = resource.getrusage(resource.RUSAGE_SELF)
cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
print(f"n={n}: {fits / dt:.4f} fits/s; host CPU {cpu / dt:.2f} cores busy on average (user {r1.ru_utime - r0.ru_utime:.2f} s, system {r1.ru_stime - r0.ru_stime:.2f} s over {dt:.2f} s)")
ctx.close()
