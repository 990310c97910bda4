"""GPU: the device-scheduled factorisation (one persistent task-queue launch, csrc/dag_kernel.inc.hpp) against the
launch-per-product path and the oracle.
  * recursion plan (HBEGP_DAG_RL=0): per output element the same sequence of MFMA accumulations as the launch path, so lml,
    gradient, alpha, K^-1 and diag(L) must be BITWISE equal -- with the K^-1 = X^T X tiles inside the queue or as a launch;
  * right-looking plan (the default): the same factor by another order of operations -- equal to the launch path to 1e-11
    of the largest entry, bitwise reproducible for any number of workgroups and across evaluation slots;
  * the oracle comparison pins everything to lml.rs:29-79."""
import math

import numpy as np
import pytest

from hbetune_rs_amd import gpr, synth
from oracle import gpr_oracle as O

pytestmark = pytest.mark.gpu


def _eval_all(X, y, theta, monkeypatch, dag, n_slots=1, reps=2, **env):
    monkeypatch.setenv("HBEGP_DAG", dag)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    prob = gpr.Problem(X, y, n_slots=n_slots)
    out = []
    for rep in range(reps):
        r = prob.lml_with_gradient(theta + 0.02 * rep)
        out.append((r, prob.results()))
    prob.close()
    return out


@pytest.mark.parametrize("n,cfg,dtype", [(200, "C2", np.float64), (256, "C1", np.float64), (300, "M", np.float64), (1100, "C3", np.float64),
                                          (1300, "M", np.float64), (1409, "C2", np.float64), (1900, "M", np.float64),  # 11, 12, 15 blocks: odd splits
                                          (2048, "M", np.float64), (600, "C5", np.float32), (1536, "C5", np.float32)])
def test_task_queue_is_bitwise_equal_to_launch_path(n, cfg, dtype, monkeypatch):
    w = synth.make_workload(cfg, n=n)
    X, y, theta = w["X"].astype(dtype), w["y"].astype(dtype), w["theta"].copy()
    if dtype == np.float32:
        theta[0] = theta[1] + math.log(0.5)
        monkeypatch.setenv("HBEGP_F32_REFINE", "0")  # the refined f32 panel solve exists as launches only: compare the plain recursion
    ref = _eval_all(X, y, theta, monkeypatch, "0")
    # any number of workgroups must give the same bits (and terminate); K^-1 tiles in the queue (default) or as a launch
    for wg, small_h, lauum in [(0, 8, 1), (3, 2, 1), (40, 0, 0)]:
        got = _eval_all(X, y, theta, monkeypatch, "1", HBEGP_DAG_RL=0, HBEGP_DAG_WG=wg, HBEGP_DAG_SMALLH=small_h, HBEGP_DAG_LAUUM=lauum,
                        HBEGP_DAG_VALIDATE=1)
        for (r0, (a0, k0, l0)), (r1, (a1, k1, l1)) in zip(ref, got):
            assert r0 is not None and r1 is not None
            assert r0[0] == r1[0] and np.array_equal(r0[1], r1[1])
            assert np.array_equal(a0, a1) and np.array_equal(k0, k1) and np.array_equal(l0, l1)


def _close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all(np.isfinite(b))) and float(np.abs(a - b).max()) <= tol * max(1.0, float(np.abs(a).max()))


@pytest.mark.parametrize("n,cfg,dtype", [(200, "C2", np.float64), (300, "M", np.float64), (1100, "C3", np.float64), (1409, "C2", np.float64),
                                          (2048, "M", np.float64), (2300, "M", np.float64), (600, "C5", np.float32), (1536, "C5", np.float32)])
def test_right_looking_plan_agrees_with_launch_path_and_is_reproducible(n, cfg, dtype, monkeypatch):
    w = synth.make_workload(cfg, n=n)
    X, y, theta = w["X"].astype(dtype), w["y"].astype(dtype), w["theta"].copy()
    tol = 1e-11
    if dtype == np.float32:
        theta[0] = theta[1] + math.log(0.5)
        monkeypatch.setenv("HBEGP_F32_REFINE", "0")
        tol = 2e-5  # two f32 orders of operations
    ref = _eval_all(X, y, theta, monkeypatch, "0")
    first = None
    for wg, near, group in [(0, 1, 32), (3, 1, 32), (40, 1, 32), (0, 2, 4)]:
        got = _eval_all(X, y, theta, monkeypatch, "1", HBEGP_DAG_RL=1, HBEGP_DAG_WG=wg, HBEGP_DAG_RL_NEAR=near, HBEGP_DAG_RL_GROUP=group,
                        HBEGP_DAG_VALIDATE=1)
        for (r0, (a0, k0, l0)), (r1, (a1, k1, l1)) in zip(ref, got):
            assert r0 is not None and r1 is not None
            assert _close(r0[0], r1[0], tol) and _close(r0[1], r1[1], tol)
            assert _close(a0, a1, tol) and _close(np.tril(k0), np.tril(k1), tol) and _close(l0, l1, tol)
        if (near, group) == (1, 32):  # same plan, another number of workgroups: same bits
            if first is None:
                first = got
            for (r0, (a0, k0, l0)), (r1, (a1, k1, l1)) in zip(first, got):
                assert r0[0] == r1[0] and np.array_equal(r0[1], r1[1]) and np.array_equal(a0, a1) and np.array_equal(k0, k1)


@pytest.mark.parametrize("rl", [0, 1])
def test_task_queue_matches_oracle(rl, monkeypatch):
    w = synth.make_workload("C2", n=700)
    X, y, theta = w["X"], w["y"], w["theta"]
    (r, (alpha, kinv, ldiag)), = _eval_all(X, y, theta, monkeypatch, "1", reps=1, HBEGP_DAG_RL=rl)
    s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
    ref = O.lml_with_gradient(X, y, s2, c, ell, 2.5)
    assert abs(r[0] - ref["lml"]) <= 1e-8 * max(1.0, abs(ref["lml"]))
    np.testing.assert_allclose(r[1], ref["grad"], rtol=0, atol=1e-8 * max(1.0, np.abs(ref["grad"]).max()))
    np.testing.assert_allclose(alpha, ref["alpha"], rtol=0, atol=1e-8 * max(1.0, np.abs(ref["alpha"]).max()))
    np.testing.assert_allclose(np.tril(kinv), np.tril(ref["k_inv"]), rtol=0, atol=1e-8 * max(1.0, np.abs(ref["k_inv"]).max()))


@pytest.mark.parametrize("rl", [0, 1])
def test_task_queue_drains_when_not_positive_definite(rl, monkeypatch):
    # lml.rs:47-50: the failing diagonal block flags the evaluation; every later task skips its work but still bumps its
    # counters, so the queue drains at once and the slot is usable again
    monkeypatch.setenv("HBEGP_DAG", "1")
    monkeypatch.setenv("HBEGP_DAG_RL", str(rl))
    rng = np.random.default_rng(5)
    X = rng.random((400, 3))
    # Copies of one row + vanishing noise: K is exactly singular, so blocks 0-1 succeed and block 2 (rows 256..) fails.  MANY
    # copies, not one: a single copy's pivot is 0 up to rounding and lands on either side (it did, when the kernel matrix's
    # exp/sqrt changed by an ulp); each further copy is another draw, 144 of them never all come out positive.
    X[256:] = X[10]
    y = rng.random(400)
    prob = gpr.Problem(X, y)
    assert prob.lml_with_gradient(np.array([math.log(1e-300), 0.0, 0.0, 0.0, 0.0])) is None
    ok = prob.lml_with_gradient(np.array([math.log(1e-2), 0.0, 0.0, 0.0, 0.0]))
    assert ok is not None and np.isfinite(ok[0]) and np.all(np.isfinite(ok[1]))


def test_concurrent_slots_share_the_chip_and_agree(monkeypatch):
    # three slots evaluated from three host threads (what a fit does): each slot's launch holds a third of the CUs
    import threading

    monkeypatch.setenv("HBEGP_DAG", "1")  # the single-slot problem below would take the launch path at this size
    w = synth.make_workload("M", n=1500)
    X, y, theta = w["X"], w["y"], w["theta"]
    prob = gpr.Problem(X, y, n_slots=3)
    res = [None] * 3

    def work(slot):
        vals = []
        for rep in range(6):
            vals.append(prob.lml_with_gradient(theta + 0.01 * ((rep + slot) % 3), slot=slot))
        res[slot] = vals

    ts = [threading.Thread(target=work, args=(s,)) for s in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    single = {k: gpr.Problem(X, y).lml_with_gradient(theta + 0.01 * k) for k in range(3)}
    for slot in range(3):
        for rep in range(6):
            k = (rep + slot) % 3
            assert res[slot][rep][0] == single[k][0] and np.array_equal(res[slot][rep][1], single[k][1])


@pytest.mark.parametrize("rl", [0, 1])
def test_concurrent_soak_against_launch_path(rl, monkeypatch):
    # tools/dag_soak.py in small: 3 slots x 40 rounds of different theta at once (odd block count), every result bit for bit
    # (recursion plan) / to 1e-11 (right-looking plan) what the launch path returns on a quiet device -- a stale operand
    # anywhere would show
    import threading

    monkeypatch.setenv("HBEGP_DAG", "1")
    monkeypatch.setenv("HBEGP_DAG_RL", str(rl))
    w = synth.make_workload("M", n=1300)
    X, y = w["X"], w["y"]
    rng = np.random.default_rng(7)
    rounds = 40
    thetas = w["theta"][None, :] + 0.15 * rng.standard_normal((3 * rounds, len(w["theta"])))
    prob = gpr.Problem(X, y, n_slots=3)
    got = [None] * len(thetas)

    def work(slot):
        for r in range(rounds):
            got[3 * r + slot] = prob.lml_with_gradient(thetas[3 * r + slot], slot=slot)

    ts = [threading.Thread(target=work, args=(s,)) for s in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    prob.close()
    monkeypatch.setenv("HBEGP_DAG", "0")
    ref = gpr.Problem(X, y)
    for i, th in enumerate(thetas):
        r = ref.lml_with_gradient(th)
        assert (r is None) == (got[i] is None)
        if r is not None and rl == 0:
            assert r[0] == got[i][0] and np.array_equal(r[1], got[i][1]), i
        elif r is not None:
            assert _close(r[0], got[i][0], 1e-11) and _close(r[1], got[i][1], 1e-11), i


@pytest.mark.parametrize("n,dag", [(128, "0"), (640, "0"), (1100, "0"), (2048, "1")])
def test_diagonal_block_helper_waves_may_start_late(n, dag, monkeypatch):
    # Root cause of round 2's run-to-run deviations (profiles/r03_leaf_race.txt): the helper waves of the
    # diagonal-block kernel re-read the pivot rows that wave 0 overwrites with L at the end of the same phase, with no barrier
    # in between -- correct only while a helper is not more than part of wave 0's elimination late.  The helpers now read a
    # copy nobody writes during the phase (LEAF_DIAG_COPY).  HBEGP_LEAF_DBG=16 holds every helper back ~3 us, longer than
    # wave 0's whole elimination: lml, gradient, alpha, K^-1 and diag(L) must not change in a single bit.  (A library built
    # without that copy fails this test at every n.)  dag = "1" at n = 2048: the task queue's diagonal block -- the called,
    # write-through instantiation every evaluation from n = 2048 on runs (default right-looking plan); the debug bit reaches it
    # through DagLaunch::leaf_dbg.
    w = synth.make_workload("M", n=n)
    X, y, theta = w["X"], w["y"], w["theta"]
    ref = _eval_all(X, y, theta, monkeypatch, dag)
    got = _eval_all(X, y, theta, monkeypatch, dag, HBEGP_LEAF_DBG=16)
    for (r0, (a0, k0, l0)), (r1, (a1, k1, l1)) in zip(ref, got):
        assert r0 is not None and r1 is not None
        assert r0[0] == r1[0] and np.array_equal(r0[1], r1[1])
        assert np.array_equal(a0, a1) and np.array_equal(k0, k1) and np.array_equal(l0, l1)
    # and both agree with the oracle (the delayed run is not merely equal to a wrong undelayed one)
    want = O.lml_with_gradient(X, y, math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:]), 2.5)
    assert abs(got[0][0][0] - want["lml"]) <= 1e-8 * max(1.0, abs(want["lml"]))
    np.testing.assert_allclose(got[0][0][1], want["grad"], rtol=0, atol=1e-8 * max(1.0, np.abs(want["grad"]).max()))


@pytest.mark.parametrize("n", [1100, 2048])
def test_split_kinv_tiles_continue_the_same_accumulation(n, monkeypatch):
    # One evaluation alone forms the top-left K^-1 tiles in two parts (HBEGP_DAG_LAUUM_SPLIT, the default for a single slot;
    # a fit's slots do not): the second part starts its MFMA chain from the first part's stored sums (DAGF_CINIT), so both
    # orders give the same bits -- `extend` after `estimate` is bit-comparable with the fit's own evaluation of that theta.
    w = synth.make_workload("M", n=n)
    X, y, theta = w["X"], w["y"], w["theta"]
    one = _eval_all(X, y, theta, monkeypatch, "1", HBEGP_DAG_LAUUM_SPLIT=0)
    two = _eval_all(X, y, theta, monkeypatch, "1", HBEGP_DAG_LAUUM_SPLIT=1)
    for (r0, (a0, k0, l0)), (r1, (a1, k1, l1)) in zip(one, two):
        assert r0[0] == r1[0] and np.array_equal(r0[1], r1[1])
        assert np.array_equal(a0, a1) and np.array_equal(k0, k1) and np.array_equal(l0, l1)


def _bits_equal(a, b):
    (ra, (aa, ka, la)), (rb, (ab, kb, lb)) = a, b
    return ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(aa, ab) and np.array_equal(np.tril(ka), np.tril(kb)) and np.array_equal(la, lb)


@pytest.mark.parametrize("n,cfg,dtype", [(300, "M", np.float64), (800, "C2", np.float64), (1500, "M", np.float64), (2700, "M", np.float64), (1500, "C5", np.float32)])
def test_round4_switches_do_not_change_a_bit(n, cfg, dtype, monkeypatch):
    # Round 4 changed HOW an evaluation is driven and scheduled, not its arithmetic: (a) through pinned memory, with the gradient
    # finalised and the lml formed by the last workgroup of a launch, or with copy nodes, separate finalise kernels and a
    # stream synchronise (HBEGP_HOSTIO); (b) 32x64 one-shot tiles between two diagonal blocks or 64x64 staged tiles
    # (HBEGP_DAG_CHAIN32: the same k-ascending accumulation per element).  lml, gradient, alpha, K^-1 and diag(L) bit for bit.
    w = synth.make_workload(cfg, n=n)
    X, y, theta = w["X"].astype(dtype), w["y"].astype(dtype), w["theta"].copy()
    if dtype == np.float32:
        theta[0] = theta[1] + math.log(0.5)
    monkeypatch.delenv("HBEGP_DAG", raising=False)
    base = None
    for hostio, chain32 in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("HBEGP_HOSTIO", hostio)
        monkeypatch.setenv("HBEGP_DAG_CHAIN32", chain32)
        prob = gpr.Problem(X, y)
        got = (prob.lml_with_gradient(theta), prob.results())
        prob.close()
        base = base or got
        assert _bits_equal(base, got), (hostio, chain32)


@pytest.mark.parametrize("n", [900, 1500, 2500])
def test_row_progressive_plan_against_the_divide_and_conquer_inverse(n, monkeypatch):
    # The inverse of the factor and K^-1 row by row behind the diagonal chain (the default up to 20 blocks, csrc/dag_plan.hpp
    # rl_progressive) against the divide-and-conquer inverse: the same factor (bitwise: the Cholesky part of the plan is the same),
    # another order of operations for X = L^-1 -- equal to 1e-11 of the largest entry, and both within 1e-8 of the oracle.  In f64
    # the ranges a continued sum is cut into do not matter (DAGF_CINIT: one k-ascending chain per element): bit for bit.
    w = synth.make_workload("M", n=n)
    X, y, theta = w["X"], w["y"], w["theta"]
    monkeypatch.delenv("HBEGP_DAG", raising=False)
    monkeypatch.setenv("HBEGP_DAG_MIN_BLOCKS", "2")
    out = {}
    for name, env in (("dc", {"HBEGP_DAG_PROG": "0"}), ("prog", {"HBEGP_DAG_PROG": "1"}), ("prog3", {"HBEGP_DAG_PROG": "1", "HBEGP_DAG_PROG_RATIO": "3"})):
        for k in ("HBEGP_DAG_PROG", "HBEGP_DAG_PROG_RATIO"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        prob = gpr.Problem(X, y)
        out[name] = (prob.lml_with_gradient(theta), prob.results())
        prob.close()
    assert _bits_equal(out["prog"], out["prog3"])
    (rd, (ad, kd, ld)), (rp, (ap, kp, lp)) = out["dc"], out["prog"]
    assert np.array_equal(ld, lp)  # diag(L): the factor itself is the same computation
    assert abs(rd[0] - rp[0]) <= 1e-11 * max(1.0, abs(rd[0]))
    np.testing.assert_allclose(rp[1], rd[1], rtol=0, atol=1e-11 * max(1.0, np.abs(rd[1]).max()))
    np.testing.assert_allclose(ap, ad, rtol=0, atol=1e-11 * max(1.0, np.abs(ad).max()))
    np.testing.assert_allclose(np.tril(kp), np.tril(kd), rtol=0, atol=1e-11 * max(1.0, np.abs(kd).max()))
    s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
    ref = O.lml_with_gradient(X, y, s2, c, ell, 2.5)
    assert abs(rp[0] - ref["lml"]) <= 1e-8 * max(1.0, abs(ref["lml"]))
    np.testing.assert_allclose(np.tril(kp), np.tril(ref["k_inv"]), rtol=0, atol=1e-8 * max(1.0, np.abs(ref["k_inv"]).max()))
