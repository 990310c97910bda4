#!/usr/bin/env python3
"""Which output of an evaluation is not reproducible?  The same theta evaluated again and again on 3 slots at once and on one
slot alone; every difference in lml / gradient / alpha / K^-1 / diag(L) against the first result is reported."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hbetune_rs_amd import gpr, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
w = synth.make_workload("M", n=n)
os.environ.setdefault("HBEGP_DAG", "1")
for slots in (1, 3):
    prob = gpr.Problem(w["X"], w["y"], n_slots=slots)
    ref = None
    bad = {"lml": 0, "grad": 0, "alpha": 0, "kinv": 0, "ldiag": 0}
    lock = threading.Lock()
    def work(slot):
        global ref
        for r in range(reps):
            out = prob.lml_with_gradient(w["theta"], slot=slot)
            a, k, l = prob.results(slot=slot)
            cur = (out[0], out[1].copy(), a.copy(), np.tril(k).copy(), l.copy())
            with lock:
                if ref is None:
                    ref = cur
                else:
                    for name, x, y in zip(bad, ref, cur):
                        if not np.array_equal(np.asarray(x), np.asarray(y)):
                            bad[name] += 1
                            if bad[name] <= 2:
                                d = np.abs(np.asarray(x, dtype=float) - np.asarray(y, dtype=float))
                                print(f"  slots={slots} slot {slot} rep {r}: {name} differs, max |d| {d.max():.3e} at {np.unravel_index(np.argmax(d), d.shape) if d.ndim else ()} count {int((d > 0).sum()) if d.ndim else 1}")
    ts = [threading.Thread(target=work, args=(s,)) for s in range(slots)]
    [t.start() for t in ts]; [t.join() for t in ts]
    print(f"n={n} slots={slots}: {reps * slots} evaluations, differences: {bad}")
    prob.close()
