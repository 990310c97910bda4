#!/usr/bin/env python3
"""Where a tile task's time goes, from a task trace written by a -DDAG_STAMP_INNER build (make variant NAME=stamp DEFS=-DDAG_STAMP_INNER;
HBEGP_LIB=build/var/libhbegp_stamp.so HBEGP_DAG_TRACE=... tools/trace_eval.py): per class the medians of
  head    ready -> first stage in the LDS (the first MFMA can start)
  loop    -> last MFMA issued
  tail    -> results stored (write-through stores issued)
  publish -> counters bumped
usage: trace_inner.py <trace file>"""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
kind, depth = a[:, 1], a[:, 4]
ready, comp, pub = a[:, 7], a[:, 8], a[:, 9]
st1, st2 = a[:, 10], a[:, 11]
lo = lambda x: x & 0xffffffff
d = lambda x, y: ((lo(x) - lo(y)) & 0xffffffff) / 100.0  # us
names = {0: "gemm128x64", 1: "gemm64x64"}
print(f"{'class':22s} {'n':>5s} {'head':>7s} {'loop':>8s} {'tail':>7s} {'publish':>8s} {'busy':>8s} {'MFMA ideal':>11s}")
for k in (0, 1):
    for dep in sorted(set(depth[kind == k])):
        m = (kind == k) & (depth == dep) & (st1 != 0)
        if m.sum() < 20:
            continue
        head, loop, tail, publ = d(st1[m], ready[m]), d(st2[m], st1[m]), d(comp[m], st2[m]), (pub[m] - comp[m]) / 100.0
        ideal = (128 if k == 0 else 64) * 64 * dep * 2.0 / 0.3072e6
        print(f"{names[k] + ' k=' + str(dep):22s} {m.sum():5d} {np.median(head):7.2f} {np.median(loop):8.2f} {np.median(tail):7.2f} {np.median(publ):8.2f} "
              f"{np.median((pub[m] - ready[m]) / 100.0):8.2f} {ideal:11.2f}")
