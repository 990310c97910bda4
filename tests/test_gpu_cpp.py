"""GPU: the header-only C++ mirror (include/hbegp.hpp) compiles against libhbegp.so and passes the reference's simple
fit->predict known-answer test (predict.rs:54-99)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_mirror_smoke(tmp_path):
    exe = str(tmp_path / "test_hbegp")
    lib_dir = os.path.join(ROOT, "hbetune_rs_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_hbegp.cpp"),
                           "-L", lib_dir, "-lhbegp", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad=0 threw=1" in out.stdout


def test_cpp_header_compiles():
    # CPU: syntax/ABI check of the header against include/hbegp.h (no link, no GPU)
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_hbegp.cpp")])
