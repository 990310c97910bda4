#!/usr/bin/env python3
"""Block map of where the task-queue path and the launch path differ in W2 (X = L^-1) / W1 after one evaluation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hbetune_rs_amd import gpr, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
w = synth.make_workload("M", n=n)
mats = {}
for mode in ("0", "1"):
    os.environ["HBEGP_DAG"] = mode
    prob = gpr.Problem(w["X"], w["y"])
    prob.lml_with_gradient(w["theta"])
    mats[mode] = (prob.debug_work_matrix(1), prob.debug_work_matrix(2))
    prob.close()
nb = mats["0"][0].shape[0] // 128
for which, name in ((1, "W2 (X)"), (0, "W1 (Schur/U)")):
    a, b = mats["0"][which], mats["1"][which]
    print(name, "blocks that differ (row: block row; '#': differs, '.': equal), lower triangle:")
    for i in range(nb):
        print("  %2d " % i + "".join("#" if not np.array_equal(a[i*128:(i+1)*128, j*128:(j+1)*128], b[i*128:(i+1)*128, j*128:(j+1)*128]) else "." for j in range(i + 1)))
