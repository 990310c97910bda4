#!/usr/bin/env python3
"""Device-scheduled factorisation (HBEGP_DAG=1) against the launch-per-product path (HBEGP_DAG=0): same bits expected
for lml, gradient, alpha, K^-1, diag(L).  Usage: dag_check.py [n ...]   (env HBEGP_DAG_WG / HBEGP_DAG_SMALLH apply)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hbetune_rs_amd import gpr, synth

sizes = [int(a) for a in sys.argv[1:]] or [256, 300, 512, 1100, 2048, 4096]
dtype = np.float32 if os.environ.get("DAG_CHECK_F32") else np.float64
TOL = float(os.environ.get("DAG_CHECK_TOL", "0"))  # > 0: compare within TOL relative to the largest entry instead of bitwise (right-looking plan)
def same(a, b):
    if TOL <= 0:
        return np.array_equal(a, b)
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all(np.isfinite(b)) and np.abs(a - b).max() <= TOL * max(1.0, np.abs(a).max()))
bad = 0
for n in sizes:
    w = synth.make_workload("M", n=n)
    X, y, theta = w["X"].astype(dtype), w["y"].astype(dtype), w["theta"].copy()
    if dtype == np.float32:
        theta[0] = theta[1] + np.log(0.5)  # well-conditioned for f32
    res = {}
    for mode in ("0", "1"):
        os.environ["HBEGP_DAG"] = mode
        prob = gpr.Problem(X, y)
        out = []
        for rep in range(3):
            th = theta + 0.01 * rep
            t0 = time.perf_counter()
            r = prob.lml_with_gradient(th)
            dt = time.perf_counter() - t0
            a, kinv, ld = prob.results()
            out.append((r, a, kinv, ld, dt))
        res[mode] = out
        if n >= 2048:
            ph = prob.time_eval(theta, reps=5)
            print(f"   n={n} DAG={mode}: eval_graph_ms={ph['eval_graph_ms']:.3f} leaf={ph['leaf_ms']:.3f} gemm={ph['chol_gemm_ms']:.3f} "
                  f"dag={ph['dag_ms']:.3f} ({ph['dag_gflop']:.1f} GFLOP) lauum={ph['lauum_ms']:.3f}")
        prob.close()
    for rep in range(3):
        (r0, a0, k0, l0, _), (r1, a1, k1, l1, dt1) = res["0"][rep], res["1"][rep]
        ok = (r0 is None) == (r1 is None)
        if ok and r0 is not None:
            ok = same(r0[0], r1[0]) and same(r0[1], r1[1]) and same(a0, a1) and same(np.tril(k0), np.tril(k1)) and same(l0, l1)
        if not ok:
            bad += 1
            if r0 is not None and r1 is not None:
                print(f"n={n} rep={rep} MISMATCH lml {r0[0]!r} vs {r1[0]!r}; max|dalpha|={np.abs(a0-a1).max():.3e} max|dKinv|={np.abs(k0-k1).max():.3e} "
                      f"max|dldiag|={np.abs(l0-l1).max():.3e} nan={np.isnan(k1).sum()}")
                d = np.abs(l0 - l1); print("   first ldiag diff at", int(np.argmax(d > 0)) if (d > 0).any() else -1)
            else:
                print(f"n={n} rep={rep} MISMATCH status", r0 is None, r1 is None)
        else:
            extra = "" if TOL <= 0 or r0 is None else f"  max rel diff: lml {abs(r0[0]-r1[0])/abs(r0[0]):.1e} grad {np.abs(r0[1]-r1[1]).max()/max(1,np.abs(r0[1]).max()):.1e} alpha {np.abs(a0-a1).max()/np.abs(a0).max():.1e} Kinv {np.abs(np.tril(k0)-np.tril(k1)).max()/np.abs(k0).max():.1e}"
            print(f"n={n} rep={rep} {'identical bits' if TOL <= 0 else 'within tolerance'} (lml={None if r0 is None else r0[0]:.6f}){extra}")
print("FAILED" if bad else "ALL IDENTICAL")
sys.exit(1 if bad else 0)
