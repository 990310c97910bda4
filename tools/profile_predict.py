#!/usr/bin/env python3
"""Wall time of extend (one evaluation + model) and of batched predict calls.  Usage: profile_predict.py [cfg] [n] [m]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth  # noqa: E402
cfg = sys.argv[1] if len(sys.argv) > 1 else "M"
n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-" else None
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1600
w = synth.make_workload(cfg, n=n)
t0 = time.perf_counter(); fk = gpr.FittedKernel.extend(w["X"], w["y"], w["theta"]); t_ext = time.perf_counter() - t0
Xs = synth.candidates(cfg, m, w["d"]).astype(w["X"].dtype)
fk.predict(Xs)
out = {"cfg": cfg, "n": w["n"], "dtype": str(w["X"].dtype), "extend_ms": t_ext * 1e3}
for mm in (1, 2, 16, 32, m):
    t0 = time.perf_counter()
    for _ in range(5): fk.predict(Xs[:mm])
    out[f"predict_m{mm}_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    t0 = time.perf_counter()
    for _ in range(5): fk.predict(Xs[:mm], want_variance=False)
    out[f"predict_mean_m{mm}_ms"] = (time.perf_counter() - t0) / 5 * 1e3
# extend: full path vs incremental (prior built on all but the last `k` rows), warm workspaces
for k in (16, 100):
    prior = gpr.FittedKernel.extend(w["X"][:-k], w["y"][:-k], w["theta"])
    for name, fn in (("full", lambda: gpr.FittedKernel.extend(w["X"], w["y"], w["theta"])),
                     ("incremental", lambda: prior.extend_with(w["X"], w["y"]))):
        fn()
        t0 = time.perf_counter()
        for _ in range(3): r = fn()
        out[f"extend_{name}_k{k}_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    assert r.incremental
print(json.dumps(out))
