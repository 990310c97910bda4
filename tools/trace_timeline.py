"""Timeline analysis of a rocprofv3 kernel trace of the multi-stream fit: how much of the wall time has a big
GEMM launch in flight, how much only small launches, how much nothing (usage: trace_timeline.py <kernel_trace.csv>)."""
import csv, sys
import numpy as np

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        if "dag_kernel" in name: cls = "dag"
        elif "gemm_kernel<double, 64" in name or "gemm_kernel<double, 128" in name: cls = "gemm_big"
        elif "gemm_kernel" in name: cls = "gemm32_full" if grid >= 256 else "gemm32_small"
        elif "leaf_kernel" in name: cls = "leaf"
        else: cls = "other"
        rows.append((s, e, cls, int(r["Queue_Id"]), "kmat_kernel" in name))
rows.sort()
# restrict to the densest stretch: the fit (drop the first and last 10 % of the launches)
lo, hi = rows[len(rows) // 10][0], rows[-len(rows) // 10][1]
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
wall = hi - lo
classes = ["dag", "gemm_big", "gemm32_full", "gemm32_small", "leaf", "other"]
ev = []
for s, e, c, q, _ in rows:
    ev.append((s, 1, c)); ev.append((e, -1, c))
ev.sort()
cnt = {c: 0 for c in classes}
t_prev = lo
acc = {}
for t, d, c in ev:
    key = tuple(c2 for c2 in classes if cnt[c2] > 0)
    acc[key] = acc.get(key, 0) + (t - t_prev)
    t_prev = t
    cnt[c] += d
print("wall ms", wall / 1e6, "launches", len(rows))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:16]:
    print(f"{100 * v / wall:6.2f} %  {'+'.join(k) if k else '(idle)'}")
for c in classes:
    tot = sum(e - s for s, e, cc, q, _ in rows if cc == c)
    union = sum(v for k, v in acc.items() if c in k)
    print(f"{c:14s} sum {tot / 1e6:9.2f} ms  union {union / 1e6:9.2f} ms ({100 * union / wall:5.1f} % of wall)  n={sum(1 for r in rows if r[2] == c)}")

# per queue: gap between the end of one evaluation and the kmat launch of the next (host turnaround), and
# the distribution of dependent-launch gaps inside an evaluation
import collections
byq = collections.defaultdict(list)
for r in rows: byq[r[3]].append(r)
for q, rs in sorted(byq.items()):
    rs.sort()
    host_gaps, inner_gaps = [], []
    for a, b in zip(rs, rs[1:]):
        gap = b[0] - a[1]
        (host_gaps if b[4] else inner_gaps).append(gap)
    if not host_gaps: continue
    hg, ig = np.array(host_gaps), np.array(inner_gaps)
    print(f"queue {q}: {len(rs)} launches, evals {len(hg)}, host turnaround median {np.median(hg)/1e3:.1f} us mean {hg.mean()/1e3:.1f} us; "
          f"inner gaps median {np.median(ig)/1e3:.2f} us mean {ig.mean()/1e3:.2f} us sum/eval {ig.sum()/len(hg)/1e3:.1f} us")
