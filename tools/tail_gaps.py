#!/usr/bin/env python3
"""Per queue of a rocprofv3 kernel trace of the three-stream fit: duration of every kernel class and the gap in front of it
(end of the previous launch on the same queue -> its own start), i.e. where an evaluation's time outside the task-queue launch
goes.  usage: tail_gaps.py <kernel_trace.csv>"""
import collections, csv, re, sys
import numpy as np

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"^hbegp::", "", name).split("(")[0].split("<")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Queue_Id"])))
rows.sort()
lo, hi = rows[len(rows) // 10][0], rows[-len(rows) // 10][1]
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
byq = collections.defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
for q, rs in byq.items():
    if sum(1 for r in rs if r[2] == "kmat_kernel") < 50:
        continue
    for a, b in zip(rs, rs[1:]):
        dur[b[2]].append(b[1] - b[0])
        gap[(a[2], b[2])].append(b[0] - a[1])
print(f"{'kernel':28s} {'n':>6s} {'median us':>10s} {'mean us':>10s}")
for k, v in sorted(dur.items(), key=lambda kv: -np.sum(kv[1])):
    v = np.array(v) / 1e3
    print(f"{k:28s} {len(v):6d} {np.median(v):10.1f} {v.mean():10.1f}")
print()
print(f"{'gap: previous -> next':50s} {'n':>6s} {'median us':>10s} {'mean us':>10s} {'p90 us':>10s} {'max us':>10s}")
tot = 0.0
nev = max(1, len(dur.get("kmat_kernel", [])))
for k, v in sorted(gap.items(), key=lambda kv: -np.sum(kv[1])):
    v = np.array(v) / 1e3
    if len(v) < 20:
        continue
    tot += v.sum()
    print(f"{k[0] + ' -> ' + k[1]:50s} {len(v):6d} {np.median(v):10.1f} {v.mean():10.1f} {np.percentile(v, 90):10.1f} {v.max():10.1f}")
print(f"gaps per evaluation: {tot / nev:.1f} us; kernels other than the task-queue launch per evaluation: "
      f"{sum(np.sum(v) for k, v in dur.items() if k != 'dag_kernel') / 1e3 / nev:.1f} us" +
      (f"; task-queue launch mean {np.mean(dur['dag_kernel']) / 1e3:.1f} us" if "dag_kernel" in dur else ""))
print("launches per evaluation:", " ".join(f"{k}={len(v) / nev:.1f}" for k, v in sorted(dur.items(), key=lambda kv: -len(kv[1])) if len(v) >= nev // 2))
