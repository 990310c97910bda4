#!/usr/bin/env python3
"""Soak of the task-queue path under the fit's concurrency: 3 slots evaluate different theta at once, round after round;
every (lml, gradient) must equal what a quiet device returns for the same theta: bit for bit the single-slot task queue (same
plan, 256 workgroups), and the launch path bit for bit (recursion plan, HBEGP_DAG_RL=0) or to 1e-11 (right-looking plan).
A stale operand anywhere (missed dependency, cache visibility) shows up as a mismatch.  Usage: dag_soak.py [n] [rounds]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hbetune_rs_amd import gpr, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
w = synth.make_workload("M", n=n)
X, y, theta0 = w["X"], w["y"], w["theta"]
rng = np.random.default_rng(7)
thetas = theta0[None, :] + 0.15 * rng.standard_normal((3 * rounds, len(theta0)))
if os.environ.get("SOAK_DAG"):
    os.environ["HBEGP_DAG"] = os.environ["SOAK_DAG"]
else:
    os.environ.pop("HBEGP_DAG", None)
prob = gpr.Problem(X, y, n_slots=3)  # task queue (default for concurrent slots)
got = [None] * len(thetas)

def work(slot):
    for r in range(rounds):
        i = 3 * r + slot
        got[i] = prob.lml_with_gradient(thetas[i], slot=slot)

ts = [threading.Thread(target=work, args=(s,)) for s in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
prob.close()
bitwise_launch = os.environ.get("HBEGP_DAG_RL") == "0"
bad = 0
for mode, bitwise in (("1", True), ("0", bitwise_launch)):
    os.environ["HBEGP_DAG"] = mode
    ref = gpr.Problem(X, y)
    for i, th in enumerate(thetas):
        r = ref.lml_with_gradient(th)
        g = got[i]
        same = (r is None) == (g is None)
        if same and r is not None:
            if bitwise:
                same = r[0] == g[0] and np.array_equal(r[1], g[1])
            else:
                same = abs(r[0] - g[0]) <= 1e-11 * abs(r[0]) and np.abs(r[1] - g[1]).max() <= 1e-11 * max(1.0, np.abs(r[1]).max())
        if not same:
            bad += 1
            if bad <= 5:
                print("MISMATCH at", i, "vs", "single-slot task queue" if mode == "1" else "launch path", None if r is None else r[0], None if g is None else g[0],
                      "max |dgrad| %.3e of %.3e" % (np.abs(r[1] - g[1]).max(), np.abs(r[1]).max()) if (r and g) else "")
    ref.close()
print(f"n={n}: {len(thetas)} concurrent task-queue evaluations, {bad} mismatches against the quiet-device references")
sys.exit(1 if bad else 0)
