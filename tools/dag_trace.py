#!/usr/bin/env python3
"""Analyse a task trace of the device-scheduled factorisation (HBEGP_DAG_TRACE=<file>, written by hbegp_problem_time_eval).
Columns: idx kind row col depth nwait pulled ready computed published xcc hwid   (times in ticks of 10 ns).
Prints per task class: count, compute time (ready -> computed), publish time (computed -> published), wait (pulled -> ready)."""
import sys
import numpy as np

a = np.loadtxt(sys.argv[1], dtype=np.int64)
t0 = a[:, 6].min()
pulled, ready, comp, pub = [(a[:, c] - t0) * 0.01 for c in (6, 7, 8, 9)]  # microseconds
kind, depth, nwait = a[:, 1], a[:, 4], a[:, 5]
print(f"tasks {len(a)}  makespan {pub.max():.1f} us  CUs used {len(set(zip(a[:,10], a[:,11] & 0xffffff00)))}")
names = {0: "gemm128x64", 1: "gemm64x64", 2: "leaf", 8: "gemm32x64", 9: "gemm128x128"}
print(f"{'class':24s} {'n':>5s} {'compute':>9s} {'publish':>9s} {'wait':>9s} {'total busy us':>14s}")
for k in (2, 8, 1, 0, 9):
    for d in sorted(set(depth[kind == k])):
        m = (kind == k) & (depth == d)
        c, p, w = comp[m] - ready[m], pub[m] - comp[m], ready[m] - pulled[m]
        print(f"{names[k] + ' k=' + str(d):24s} {m.sum():5d} {np.median(c):9.2f} {np.median(p):9.2f} {np.median(w):9.2f} {(pub[m] - ready[m]).sum():14.0f}")
# where the busy time exceeds the MFMA-only ideal (one CU: 128 fp64 flop per clock at 2.4 GHz = 0.307 TFLOP/s), by class
ideal_all, busy_all = 0.0, 0.0
rows = []
for k in (8, 1, 0, 9):
    for d in sorted(set(depth[kind == k])):
        m = (kind == k) & (depth == d)
        flop = {0: 128 * 64, 1: 64 * 64, 8: 32 * 64, 9: 128 * 128}[k] * d * 2.0
        ideal = flop / 0.3072e6  # us
        b = (pub[m] - ready[m])
        rows.append((b.sum() - ideal * m.sum(), names[k], d, int(m.sum()), ideal, float(np.median(b))))
        ideal_all += ideal * m.sum(); busy_all += b.sum()
rows.sort(reverse=True)
print(f"tile tasks: busy {busy_all:.0f} us, MFMA-only ideal {ideal_all:.0f} us ({100 * ideal_all / busy_all:.1f} %); largest excess by class:")
for ex, nm, d, cnt, ideal, med in rows[:12]:
    print(f"   {nm} k={d}: {cnt} tasks, median busy {med:.2f} us vs ideal {ideal:.2f} us, excess {ex:.0f} us")
busy = (pub - ready).sum()
held = (pub - pulled).sum()
print(f"sum busy {busy:.0f} us, sum held (incl. waiting) {held:.0f} us, makespan x CUs = {pub.max() * len(set(zip(a[:,10], a[:,11] & 0xffffff00))):.0f}")
# leaf chain: time between consecutive leaves becoming ready
lm = kind == 2
order = np.argsort(ready[lm])
lr, lp = ready[lm][order], pub[lm][order]
gaps = lr[1:] - lp[:-1]
print("leaf chain: leaf busy median %.1f us; gap leaf(k) published -> leaf(k+1) ready: median %.1f max %.1f sum %.0f us" % (np.median(lp - lr), np.median(gaps), gaps.max(), gaps.sum()))
print("gaps:", " ".join(f"{g:.0f}" for g in gaps))

# hand-off between consecutive tasks of one workgroup (a workgroup stays on its CU): end of compute of task a -> task b running
cu = a[:, 10] * (1 << 32) + (a[:, 11] & 0xffffff00)
gaps_all, gaps_early, gaps_spin, n_early = [], [], [], 0
for c in set(cu):
    m = np.nonzero(cu == c)[0]
    m = m[np.argsort(ready[m])]
    for x, y in zip(m[:-1], m[1:]):
        gap = ready[y] - comp[x]
        early = pulled[y] < pub[x]
        gaps_all.append(gap)
        (gaps_early if early else gaps_spin).append(gap)
ga, ge, gs = np.array(gaps_all), np.array(gaps_early), np.array(gaps_spin)
print(f"hand-offs: {len(ga)}; computed(a) -> ready(b): median {np.median(ga):.2f} us, mean {ga.mean():.2f}, sum {ga.sum():.0f} us")
if len(ge):
    print(f"   staged by the early look: {len(ge)} ({100.0 * len(ge) / len(ga):.0f} %), median {np.median(ge):.2f} us, mean {ge.mean():.2f}")
if len(gs):
    print(f"   waited behind the barrier: {len(gs)}, median {np.median(gs):.2f} us, mean {gs.mean():.2f}, of which <= 5 us: {int((gs <= 5).sum())}")
