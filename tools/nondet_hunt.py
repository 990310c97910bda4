#!/usr/bin/env python3
"""Bisect of the run-to-run non-reproducibility under concurrent evaluation streams.

Three slots evaluate a DIFFERENT theta on every evaluation at once (the fit's concurrency); every (lml, gradient) is
compared bit for bit with what a quiet device (one slot, one stream, same library, same path) returns for the same theta.
The first few deviating evaluations are frozen on the spot -- W1 (inputs of the diagonal blocks: their final Schur
complements stay in place), W2 (X = L^-1), W3 (the factor, right-looking task queue), the raw K^-1 tiles, alpha, diag(L) --
and diffed afterwards against the quiet device's matrices for the same theta, 128-block by 128-block, to name the first
step of the factorisation whose OUTPUT deviates although its INPUT does not.  For every deviating entry the tool also checks
whether it carries the value of the evaluation that ran on that slot just before (a stale operand) .

usage: nondet_hunt.py [n=2048] [rounds=2000] [mode=rand|walk|slotconst|same] [maxdump=3]
env:   HBEGP_DAG=0 launch-per-product path; HBEGP_LIB=<variant build>; HUNT_OUT=<json summary>
"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hbetune_rs_amd import gpr, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
mode = sys.argv[3] if len(sys.argv) > 3 else "rand"
maxdump = int(sys.argv[4]) if len(sys.argv) > 4 else 3
NS = 3

w = synth.make_workload("M", n=n)
X, y, theta0 = w["X"], w["y"], w["theta"]
p = len(theta0)
rng = np.random.default_rng(11)
N = NS * rounds
if mode == "rand":      # unrelated theta from one evaluation to the next: a stale operand gives a gross error
    thetas = theta0[None, :] + 0.15 * rng.standard_normal((N, p))
elif mode == "walk":    # like an optimiser late in its run: tiny steps, a stale operand gives a tiny error
    thetas = np.empty((N, p))
    for s in range(NS):
        th = theta0 + 0.1 * rng.standard_normal(p)
        for r in range(rounds):
            th = th + 1e-4 * rng.standard_normal(p)
            thetas[NS * r + s] = th
elif mode == "slotconst":  # every slot its own theta, never changing: only cross-slot contamination can show
    base = theta0[None, :] + 0.15 * rng.standard_normal((NS, p))
    thetas = np.tile(base, (rounds, 1))
elif mode == "same":
    thetas = np.tile(theta0[None, :], (N, 1))
else:
    raise SystemExit("mode?")


def grab(prob, slot):
    out = {"W1": prob.debug_work_matrix(1, slot=slot), "W2": prob.debug_work_matrix(2, slot=slot),
           "Kraw": prob.debug_work_matrix(4, slot=slot)}
    try:
        out["W3"] = prob.debug_work_matrix(3, slot=slot)
    except Exception:
        pass
    a, _, l = prob.results(want_kinv=False, slot=slot)
    out["alpha"], out["ldiag"] = a, l
    return out


# ---- quiet reference ---------------------------------------------------------------------------------------------
t0 = time.time()
ref_prob = gpr.Problem(X, y, n_slots=1)
uniq = {}
ref = [None] * N
for i in range(N):
    key = thetas[i].tobytes()
    if key not in uniq:
        uniq[key] = ref_prob.lml_with_gradient(thetas[i])
    ref[i] = uniq[key]
t_ref = time.time() - t0

# ---- concurrent run -----------------------------------------------------------------------------------------------
prob = gpr.Problem(X, y, n_slots=NS)
bad = []          # (i, slot, dlml_rel, dgrad_rel)
dumps = []        # (i, slot, prev_i, matrices)
lock = threading.Lock()


def same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return a[0] == b[0] and np.array_equal(a[1], b[1])


def work(slot):
    for r in range(rounds):
        i = NS * r + slot
        got = prob.lml_with_gradient(thetas[i], slot=slot)
        if not same(got, ref[i]):
            with lock:
                if got is None or ref[i] is None:
                    bad.append((i, slot, float("nan"), float("nan")))
                else:
                    bad.append((i, slot, abs(got[0] - ref[i][0]) / abs(ref[i][0]),
                                float(np.abs(got[1] - ref[i][1]).max() / max(1.0, np.abs(ref[i][1]).max()))))
                take = len(dumps) < maxdump
                if take:
                    dumps.append(None)
                    k = len(dumps) - 1
            if take:
                dumps[k] = (i, slot, i - NS if r > 0 else -1, grab(prob, slot))


t0 = time.time()
ts = [threading.Thread(target=work, args=(s,)) for s in range(NS)]
[t.start() for t in ts]
[t.join() for t in ts]
t_conc = time.time() - t0
prob.close()

print(f"n={n} mode={mode} lib={os.environ.get('HBEGP_LIB', 'default')} HBEGP_DAG={os.environ.get('HBEGP_DAG', 'unset')}: "
      f"{N} concurrent evaluations ({t_conc:.1f} s; quiet reference {t_ref:.1f} s), {len(bad)} deviate from the quiet device")
for b in bad[:12]:
    print(f"   eval {b[0]} slot {b[1]}: |dlml|/|lml| {b[2]:.3e}  max|dgrad|/scale {b[3]:.3e}")

# ---- diff of the frozen evaluations ------------------------------------------------------------------------------------
NBK = 128
summary = {"n": n, "mode": mode, "evals": N, "deviating": len(bad), "lib": os.environ.get("HBEGP_LIB", "default"),
           "HBEGP_DAG": os.environ.get("HBEGP_DAG", "unset"), "first": [list(map(float, b)) for b in bad[:12]], "dumps": []}


def block_map(a, b, lower_only=True):
    nb = a.shape[0] // NBK
    m = np.zeros((nb, nb), dtype=int)
    d = a != b
    d &= ~(np.isnan(a) & np.isnan(b))
    for i in range(nb):
        for j in range(nb):
            if lower_only and j > i:
                continue
            blk = d[i * NBK:(i + 1) * NBK, j * NBK:(j + 1) * NBK]
            if lower_only and i == j:
                blk = np.tril(blk)
            m[i, j] = int(blk.sum())
    return m


for (i, slot, prev_i, got) in [d for d in dumps if d is not None]:
    if prev_i >= 0:
        ref_prob.lml_with_gradient(thetas[prev_i])
        prev = grab(ref_prob, 0)
    else:
        prev = None
    ref_prob.lml_with_gradient(thetas[i])
    want = grab(ref_prob, 0)
    print(f"--- frozen evaluation {i} (slot {slot}, previous on that slot: {prev_i})")
    rec = {"eval": i, "slot": slot, "prev": prev_i, "matrices": {}}
    for name in ("W1", "W3", "W2", "Kraw"):
        if name not in got or name not in want:
            continue
        m = block_map(got[name], want[name])
        nz = np.argwhere(m > 0)
        line = f"   {name}: {len(nz)} lower 128-blocks differ"
        info = {"blocks_differing": int(len(nz))}
        if len(nz):
            first = sorted(map(tuple, nz), key=lambda t: (t[1], t[0]))[:6]   # column-major: the factorisation's order
            line += f"; first by column: {first}; entries {[int(m[a, b]) for a, b in first]}"
            info["first_by_column"] = [list(map(int, f)) for f in first]
            # how large, and are the deviating entries the previous evaluation's values?
            d = (got[name] != want[name]) & np.tril(np.ones_like(got[name], dtype=bool))
            rel = np.abs(got[name][d] - want[name][d]) / np.maximum(np.abs(want[name][d]), 1e-300)
            line += f"; rel dev median {np.median(rel):.2e} max {rel.max():.2e}"
            info["rel_dev_median"], info["rel_dev_max"] = float(np.median(rel)), float(rel.max())
            if prev is not None and name in prev:
                stale = got[name][d] == prev[name][d]
                line += f"; {int(stale.sum())} of {int(d.sum())} deviating entries carry the PREVIOUS evaluation's value"
                info["stale_entries"], info["deviating_entries"] = int(stale.sum()), int(d.sum())
        print(line)
        rec["matrices"][name] = info
    # per diagonal block: input (W1 diag block, lower) vs outputs (ldiag, X_kk in W2)
    nb = got["W1"].shape[0] // NBK
    rows = []
    for k in range(nb):
        sl = slice(k * NBK, (k + 1) * NBK)
        in_same = np.array_equal(np.tril(got["W1"][sl, sl]), np.tril(want["W1"][sl, sl]))
        ld_same = np.array_equal(got["ldiag"][k * NBK:min(n, (k + 1) * NBK)], want["ldiag"][k * NBK:min(n, (k + 1) * NBK)])
        x_same = np.array_equal(np.tril(got["W2"][sl, sl]), np.tril(want["W2"][sl, sl]))
        rows.append((k, in_same, ld_same, x_same))
    culprit = [r for r in rows if r[1] and not (r[2] and r[3])]
    print("   diagonal blocks (k: input same / diag(L) same / X_kk same): " +
          " ".join(f"{k}:{'Y' if a else 'n'}{'Y' if b else 'n'}{'Y' if c else 'n'}" for k, a, b, c in rows))
    print(f"   diagonal blocks whose OUTPUT deviates although their INPUT is bitwise right: {[r[0] for r in culprit]}")
    rec["leaf_rows"] = [[int(k), bool(a), bool(b), bool(c)] for k, a, b, c in rows]
    rec["leaf_culprits"] = [int(r[0]) for r in culprit]
    for k, a, b, c in rows:
        if a and not c:  # where inside X_kk?  16x16 sub-blocks
            sl = slice(k * NBK, (k + 1) * NBK)
            d = np.tril(got["W2"][sl, sl] != want["W2"][sl, sl])
            sub = d.reshape(8, 16, 8, 16).sum(axis=(1, 3))
            print(f"   X_{k}{k}: deviating entries per 16x16 sub-block (rows = block row):\n" + "\n".join("      " + " ".join(f"{v:3d}" for v in row) for row in sub))
            dl = got["ldiag"][k * NBK:(k + 1) * NBK] != want["ldiag"][k * NBK:(k + 1) * NBK]
            print(f"   diag(L) of block {k}: deviating positions {np.nonzero(dl)[0].tolist()[:40]}")
            rec.setdefault("xkk_sub", {})[str(k)] = sub.tolist()
            rec.setdefault("ldiag_pos", {})[str(k)] = np.nonzero(dl)[0].tolist()
            break
    a_same = np.array_equal(got["alpha"], want["alpha"])
    print(f"   alpha same: {a_same}")
    summary["dumps"].append(rec)

ref_prob.close()
if os.environ.get("HUNT_OUT"):
    with open(os.environ["HUNT_OUT"], "w") as f:
        json.dump(summary, f, indent=1)
