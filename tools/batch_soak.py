#!/usr/bin/env python3
"""Soak of the small-fit batcher (csrc/hbegp.cpp: SmallBatcher): T host threads fit problems of up to 128 rows of different
shapes at random moments (random pauses, so that batches form, split and overlap in every way), mixed with a few fits of the
general path; every result is compared bit for bit with the same fit alone.  Usage: batch_soak.py [threads [fits_per_thread]]"""
import os
import random
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
F = int(sys.argv[2]) if len(sys.argv) > 2 else 60
shapes = [("C1", 24, 12, True), ("M", 64, 20, True), ("C2", 100, 15, False), ("M", 128, 25, True), ("C3", 90, 30, True), ("C1", 64, 8, False),
          ("C2", 300, 6, True)]  # (the last one: launch path, not batched)
ctx = gpr.Context(device_ids=[0])
jobs = []
for i, (cfg, n, maxeval, fixed) in enumerate(shapes):
    w = synth.make_workload(cfg, n=n)
    jobs.append((w, synth.restart_points(cfg, w["lo"], w["hi"], 1 + i % 3), maxeval, fixed))


def fit(job):
    w, starts, maxeval, fixed = job
    fk = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], starts, ctx=ctx, maxeval=maxeval, fixed_work=fixed)
    alpha, kinv = fk.arrays()
    mean, var, _ = fk.predict(w["X"][:3])
    out = (np.array([fk.lml, fk.n_evals]), fk.theta.copy(), alpha, kinv, mean, var)
    fk.release()
    return out


solo = [fit(j) for j in jobs]
bad, errors, done = [], [], [0]
lock = threading.Lock()


def work(seed):
    rng = random.Random(seed)
    try:
        for _ in range(F):
            i = rng.randrange(len(jobs) - 1) if rng.random() < 0.93 else len(jobs) - 1
            r = fit(jobs[i])
            if not all(np.array_equal(a, b) for a, b in zip(r, solo[i])):
                with lock:
                    bad.append((seed, i))
            with lock:
                done[0] += 1
            if rng.random() < 0.5:
                time.sleep(rng.random() * 0.004)
    except Exception as e:  # noqa: BLE001
        with lock:
            errors.append(repr(e))


t0 = time.perf_counter()
ts = [threading.Thread(target=work, args=(s,)) for s in range(T)]
[t.start() for t in ts]
[t.join() for t in ts]
dt = time.perf_counter() - t0
ctx.close()
print(f"batch soak: {T} threads x {F} fits of {len(jobs)} shapes in {dt:.1f} s ({done[0] / dt:.0f} fits/s): {len(bad)} deviations from the solo fit, {len(errors)} errors {errors[:3]}")
sys.exit(1 if bad or errors else 0)
