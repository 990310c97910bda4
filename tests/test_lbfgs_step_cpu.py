"""CPU: the resumable L-BFGS (csrc/lbfgs_step.hpp -- what the host path runs; the persistent fit kernel runs a wave-wide
transcription of it, pinned on the GPU by tests/test_gpu_fit.py::test_device_optimiser_follows_the_host_state_machine) evaluates
exactly the points of the loop form it was derived from (lbfgsb_minimize_loops), bit for bit: 14 objectives incl. bounds, failing
evaluations, fixed work, tiny budgets, 66 dimensions (tests/cpp/test_lbfgs_step.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_resumable_lbfgs_replays_the_loop_form(tmp_path):
    exe = str(tmp_path / "test_lbfgs_step")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "test_lbfgs_step.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "DIFFERENT" not in out.stdout and out.stdout.count("same") == 14
