"""CPU: the task queue of the device-scheduled factorisation (csrc/dag_plan.hpp) is sound for every shape the engine
builds -- queue order topological (so any number of resident workgroups makes progress: no deadlock by construction),
every wait for a full count, no unordered access to a tile of W1/W2 -- and the checker itself catches broken plans."""
import ctypes as C
import os

import pytest

from hbetune_rs_amd import _lib


def plan(nb, bk=16, small_h=8, nwg=85, fine=1):
    lib = _lib.load()
    nt, nc, nl = C.c_int(), C.c_int(), C.c_int()
    gf, cr, sm = C.c_double(), C.c_double(), C.c_double()
    err = C.create_string_buffer(400)
    rc = lib.hbegp_debug_dag_plan(nb, bk, small_h, nwg, fine, C.byref(nt), C.byref(nc), C.byref(nl), C.byref(gf), C.byref(cr),
                                  C.byref(sm), err, 400)
    return rc, dict(ntasks=nt.value, ncounters=nc.value, nleaf=nl.value, gflop=gf.value, crit_us=cr.value, sim_us=sm.value,
                    err=err.value.decode())


@pytest.mark.parametrize("nb", list(range(1, 20)) + [24, 31, 32, 33, 48, 64])
def test_plans_are_sound_for_every_block_count(nb):
    ref = None
    for bk in (16, 32):
        for small_h, nwg, fine in [(8, 85, 1), (2, 256, 1), (0, 1, 1), (64, 7, 1), (8, 0, 1), (8, 85, 0), (2, 0, 0)]:
            rc, info = plan(nb, bk, small_h, nwg, fine)
            assert rc == 0, (nb, bk, small_h, nwg, fine, info["err"])
            assert info["nleaf"] == nb
            ref = ref or info["gflop"]
            assert info["gflop"] == pytest.approx(ref, rel=1e-12)  # the work does not depend on tiling, order or dependencies
            if nwg > 0 and nb > 1:
                assert info["crit_us"] > 0 and info["sim_us"] >= info["crit_us"] - 1e-6


def test_flops_match_potrf_plus_trtri():
    # n = 4096: the recursion carries n^3/3 (Cholesky) + n^3/3 (inverse of the factor) minus what the half-counted
    # triangles save; LAUUM (the last third, 22.9 GFLOP) is a separate launch.  DESIGN.md quotes 68.7 - 22.95 = 45.8.
    rc, info = plan(32)
    assert rc == 0 and info["gflop"] == pytest.approx(45.77, abs=0.02)


@pytest.mark.parametrize("nb", [1, 2, 3, 5, 8, 11, 16, 24, 32, 33, 64])
def test_plans_with_the_inverse_tiles_are_sound(nb):
    # fine bit 2: K^-1 = X^T X as tile tasks behind the recursion (the default form of an evaluation)
    for bk in (16, 32):
        for small_h, nwg in [(8, 85), (2, 256), (8, 0), (0, 3)]:
            rc, info = plan(nb, bk, small_h, nwg, fine=1 | 4)
            assert rc == 0, (nb, bk, small_h, nwg, info["err"])


def test_flops_with_the_inverse_tiles_are_potrf_plus_potri():
    # n^3 = 68.7 GFLOP at n = 4096, of which lauum 22.95 (as Problem::op_gflop counts the launch it replaces)
    rc, info = plan(32, fine=1 | 4)
    assert rc == 0 and info["gflop"] == pytest.approx(45.77 + 22.95, abs=0.03)


@pytest.mark.parametrize("nb", list(range(1, 14)) + [16, 17, 24, 31, 32, 33, 48, 64, 80])
def test_right_looking_plans_are_sound_for_every_block_count(nb):
    # fine bit 3: the default plan (right-looking tile Cholesky + divide-and-conquer inverse), with and without the K^-1 tiles
    ref = None
    for bk in (16, 32):
        for small_h, nwg, fine in [(4, 96, 1 | 4 | 8), (4, 256, 1 | 4 | 8), (8, 0, 1 | 8), (0, 3, 1 | 4 | 8), (4, 85, 4 | 8)]:
            rc, info = plan(nb, bk, small_h, nwg, fine)
            assert rc == 0, (nb, bk, small_h, nwg, fine, info["err"])
            assert info["nleaf"] == nb
            if fine & 4:
                ref = ref or info["gflop"]
                assert info["gflop"] == pytest.approx(ref, rel=1e-12)
    # same algorithmic work as the recursion plan, whatever the order of operations
    rc, rec = plan(nb, 16, 8, 85, 1 | 4)
    assert rc == 0 and (ref is None or rec["gflop"] == pytest.approx(ref, rel=1e-12))


@pytest.mark.parametrize("nb", list(range(1, 14)) + [16, 17, 20, 24, 31, 32, 33, 48])
def test_row_progressive_plans_are_sound_for_every_block_count(nb):
    # fine bit 4: right-looking plan whose inverse of the factor and K^-1 follow the diagonal chain row by row (round 4; the
    # default up to 20 blocks): same flops as the divide and conquer, sound for every ordering
    _, dc = plan(nb, 16, 4, 96, 1 | 4 | 8)
    for bk in (16, 32):
        for small_h, nwg, fine in [(4, 96, 1 | 4 | 8 | 16), (4, 256, 1 | 4 | 8 | 16), (8, 0, 1 | 8 | 16), (0, 3, 1 | 4 | 8 | 16), (4, 85, 4 | 8 | 16)]:
            rc, info = plan(nb, bk, small_h, nwg, fine)
            assert rc == 0, (nb, bk, small_h, nwg, fine, info["err"])
            assert info["nleaf"] == nb
            if fine & 4:
                assert info["gflop"] == pytest.approx(dc["gflop"], rel=1e-12)


def test_row_progressive_plan_leaves_little_behind_the_last_diagonal_block():
    # n = 2048 on 256 workgroups: with the divide-and-conquer inverse a third of the flops can only start when the chain of
    # diagonal blocks has ended; row by row the critical path and the simulated makespan are shorter
    _, dc = plan(16, small_h=4, nwg=256, fine=1 | 4 | 8)
    _, pr = plan(16, small_h=4, nwg=256, fine=1 | 4 | 8 | 16)
    assert pr["crit_us"] < 0.9 * dc["crit_us"] and pr["sim_us"] < 0.9 * dc["sim_us"]


@pytest.mark.parametrize("fault", ["drop:40", "drop:700", "drop:2000", "drop:2900", "move:2500:10", "move:900:100"])
def test_checker_rejects_broken_row_progressive_plans(fault, monkeypatch):
    monkeypatch.setenv("HBEGP_DAG_TEST_FAULT", fault)
    rc, info = plan(16, small_h=4, nwg=96, fine=1 | 4 | 8 | 16)
    assert rc == _lib.EINVAL and info["err"], fault


def test_right_looking_plan_shortens_the_critical_path():
    # n = 4096: between two diagonal blocks the recursion has a product as deep as the node is wide, the right-looking plan
    # two 128-deep tiles (2.83 -> 2.03 ms simulated, 2.89 -> 2.19 ms measured for one evaluation)
    _, rec = plan(32, nwg=256, fine=1 | 4)
    _, rl = plan(32, small_h=4, nwg=256, fine=1 | 4 | 8)
    assert rl["crit_us"] < 0.8 * rec["crit_us"] and rl["sim_us"] < 0.85 * rec["sim_us"]
    assert rl["gflop"] == pytest.approx(68.72, abs=0.02)


@pytest.mark.parametrize("fault", ["drop:40", "drop:700", "drop:3000", "drop:7000", "move:5000:10", "move:900:100"])
def test_checker_rejects_broken_right_looking_plans(fault, monkeypatch):
    monkeypatch.setenv("HBEGP_DAG_TEST_FAULT", fault)
    rc, info = plan(32, small_h=4, nwg=96, fine=1 | 4 | 8)
    assert rc == _lib.EINVAL and info["err"], fault


def test_look_ahead_shortens_the_simulated_schedule():
    _, fine = plan(32, nwg=85, fine=1)
    _, coarse = plan(32, nwg=85, fine=0)
    assert fine["sim_us"] < 0.85 * coarse["sim_us"]


@pytest.mark.parametrize("fault", ["drop:40", "drop:200", "drop:900", "drop:2500", "move:3000:10", "move:500:100"])
def test_checker_rejects_broken_plans(fault, monkeypatch):
    monkeypatch.setenv("HBEGP_DAG_TEST_FAULT", fault)
    rc, info = plan(32)
    assert rc == _lib.EINVAL and info["err"], fault


def test_plan_builder_and_optimiser_are_clean_under_sanitizers(tmp_path):
    # host code of the hot path (csrc/dag_plan.hpp: both plans, 1..40 blocks, validator; csrc/lbfgsb.hpp: gradmin.rs:75-101 KAT)
    # compiled with AddressSanitizer + UndefinedBehaviorSanitizer -- sanitizers run on the CPU build only
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "san_plan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", os.path.join(root, "csrc"),
                           "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "tools", "san_plan.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "0 problems" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("nb", [2, 3, 5, 8, 16, 24, 32, 33, 48])
def test_plans_with_128x128_tiles_are_sound(nb, monkeypatch):
    # HBEGP_DAG_BIG128: the deep products without beta = 1 as 128x128 tiles (round 4; the default for multi-slot problems from 32
    # blocks on): fewer tasks, the same flops, sound in every plan -- and never a beta = 1 task of that kind (the checker rejects one)
    _, base = plan(nb, 16, 4, 96, 1 | 4 | 8)
    monkeypatch.setenv("HBEGP_DAG_BIG128", "1")
    for bk in (16, 32):
        for small_h, nwg, fine in [(4, 96, 1 | 4 | 8), (4, 256, 1 | 4 | 8 | 16), (8, 85, 1 | 4), (0, 3, 1 | 4 | 8)]:
            rc, info = plan(nb, bk, small_h, nwg, fine)
            assert rc == 0, (nb, bk, small_h, nwg, fine, info["err"])
            assert info["gflop"] == pytest.approx(base["gflop"], rel=1e-12)
    if nb >= 8:
        _, big = plan(nb, 16, 4, 96, 1 | 4 | 8)
        assert big["ntasks"] < base["ntasks"]
