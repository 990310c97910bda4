#!/usr/bin/env python3
"""Where do the loads of the 16 panel entries a[0..15] sit inside the diagonal block's elimination chain?

usage: leaf_isa_window.py <device .s file> [<function substring> ...]
(make the .s with: hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S [-DLEAF_DIAG_COPY=0] [-DDAG_LEAF_NOINLINE=1] csrc/kernels.hip)

For every function that contains an elimination (16 consecutive v_rsq_f64, one per pivot) the script prints, per ds_read of
panel data issued after the first pivot, the index of the last pivot whose v_rsq_f64 precedes it, and the instruction counts
first pivot -> that load -> first ds_write of the results.  A load that sits behind pivot k of 16 is a load the helper waves
issue roughly k/16 of the way through THEIR elimination: if wave 0 is ahead by the remaining (16-k)/16 of the chain it has
already overwritten those pivot rows with L (round 2's non-reproducibility; fixed by LEAF_DIAG_COPY)."""
import re
import sys

path = sys.argv[1]
want = sys.argv[2:] or ["leaf_kernelIdd", "dag_kernelIdLi0E", "dag_leaf_taskId"]
lines = open(path).read().split("\n")
starts = [(i, l[:-1].split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for idx, (i0, name) in enumerate(starts):
    if not any(w in name for w in want):
        continue
    i1 = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
    body = [(i, l.strip()) for i, l in enumerate(lines[i0:i1], i0) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    rsq = [k for k, (_, l) in enumerate(body) if l.startswith("v_rsq_f64")]
    # the elimination = the first run of 16 rsq with no barrier in between
    for a in range(len(rsq) - 15):
        seg = body[rsq[a]:rsq[a + 15] + 1]
        if not any(l.startswith("s_barrier") for _, l in seg):
            break
    else:
        continue
    first, last = rsq[a], rsq[a + 15]
    wr = next(k for k in range(last, len(body)) if body[k][1].startswith("ds_write"))
    print(f"{name}: elimination = {wr - first} instructions from the first pivot to the first result store")
    for k in range(first, wr):
        l = body[k][1]
        if l.startswith("ds_read"):
            piv = sum(1 for r in rsq[a:a + 16] if r < k)
            print(f"   {l:<44s} behind pivot {piv:2d} of 16, {k - first:4d} instructions in ({100.0 * (k - first) / (wr - first):.0f} % of the chain)")
