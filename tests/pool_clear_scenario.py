"""The scenario of test_fresh_pool_blocks_are_cleared_before_their_first_writer (tests/test_gpu_parity.py): a model built from
fresh pool blocks with a long null-stream fill queued in front of every clear, then predict.  Imported by the test for the
shipped clear; run as a script (python tests/pool_clear_scenario.py float64|float32) in a FRESH process for the old clear --
whether an unsynchronised null-stream clear lands on top of the model depends on which hardware queue the model's stream shares
(HIP maps streams onto a few queues in creation order; a stream that shares the null stream's queue runs behind it and is
safe by accident), and only a fresh process has a fixed creation order.  Prints one JSON line."""
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def scenario(dtype):
    from hbetune_rs_amd import gpr, synth
    from oracle import gpr_oracle as O

    n, d = 589, 5  # launch path: five 128-blocks, no other null-stream synchronisation between the pool and the first launch
    w = synth.make_workload("C2", n=n)
    rng = np.random.default_rng(589)
    X = rng.random((n, d)).astype(dtype)
    y = w["y"][:n].astype(dtype)
    theta = np.concatenate([[math.log(0.05), 0.0], np.log(np.full(d, 0.6))])
    s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
    Xs = rng.random((5, d)).astype(dtype)
    ref = O.extend(X.astype(np.float64), y.astype(np.float64), s2, c, ell, 2.5)
    rm, rv, _ = O.predict(Xs.astype(np.float64), X.astype(np.float64), ref["alpha"], ref["k_inv"], c, ell, 2.5)

    def run():
        fk = gpr.FittedKernel.extend(X, y, theta)
        try:
            # an unsynchronised clear is queued behind ~3 ms of null-stream fill per work matrix (six of them): by now it has
            # landed on top of the model's arrays, which were filled the moment they were handed out
            time.sleep(0.3)
            mean, var, _ = fk.predict(Xs)
            alpha, _ = fk.arrays()
        finally:
            fk.release()
        return (float(np.max(np.abs(mean - rm))) / max(1.0, float(np.abs(rm).max())), float(np.max(np.abs(var - np.maximum(rv, 0)))) / c,
                float(np.max(np.abs(alpha - ref["alpha"]))) / max(1.0, float(np.abs(ref["alpha"]).max())))

    return run


if __name__ == "__main__":
    from hbetune_rs_amd import gpr

    try:
        out = {"deviations": list(scenario(np.dtype(sys.argv[1]).type)())}
    except gpr.HbegpError as e:  # e.g. "not positive definite": the kernel matrix was zeroed under the factorisation
        out = {"error": str(e)}
    print(json.dumps(out), flush=True)
