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


@pytest.mark.gpu
def test_native_threads_fit_side_by_side_through_the_c_abi(tmp_path):
    # What a Rust / C++ host does: std::threads (no interpreter lock staggering them) calling hbegp_fit_f64 / hbegp_predict_f64 /
    # hbegp_model_release on ONE context at the same instant -- the small-fit batcher, the pinned per-run words, the host-side
    # turn-taking.  tools/concurrent_fits_native.cpp compares every fit with the same fit alone, bit for bit.
    exe = str(tmp_path / "concurrent_fits_native")
    wl = str(tmp_path / "wl.bin")
    lib_dir = os.path.join(ROOT, "hbetune_rs_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "concurrent_fits_native.cpp"),
                           "-L", lib_dir, "-lhbegp", "-lpthread", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    import sys
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dump_workload.py"), "100", wl])
    out = subprocess.run([exe, wl, "2", "6", "16"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("native threads")]
    assert len(lines) == 3, out.stdout + out.stderr
    for l in lines:
        assert l.rstrip().endswith("fits that differ from the solo fit: 0"), l
    print("\n".join(lines))
