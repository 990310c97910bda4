"""CPU-only checks of the boundary: the C-ABI library loads, exports every symbol include/hbegp.h declares,
refuses to run without a GPU (no silent CPU fallback), and its host-side optimiser honours the reference's KAT."""
import json
import os
import re

import numpy as np
import pytest

from hbetune_rs_amd import _lib, gpr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "hbegp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hbegp_[a-z0-9_]+)\s*\(", text)) - {"hbegp_objective_fn"})


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/hbegp.h but not exported by libhbegp.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.hbegp_version() == 200


def test_fit_options_carry_their_size():
    # hbegp.h: the options struct starts with struct_size so that it can grow at its end; the ctypes mirror has the header's
    # layout (size_t, 4 ints, 7 pointers) and a struct whose size was never set is refused before anything else is looked at
    import ctypes as C

    assert C.sizeof(_lib.FitOptions) == 8 + 4 * 4 + 7 * 8
    assert _lib.FitOptions().struct_size == C.sizeof(_lib.FitOptions)
    text = open(os.path.join(ROOT, "include", "hbegp.h")).read()
    body = re.search(r"typedef struct hbegp_fit_options \{(.*?)\} hbegp_fit_options;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b(\w+);", body)
    assert names == [f[0] for f in _lib.FitOptions._fields_]
    lib = _lib.load()
    bad = _lib.FitOptions()
    bad.struct_size = 0
    x = np.zeros(4)
    rc = lib.hbegp_fit_f64(None, _lib.dptr(x), _lib.dptr(x), 4, 1, 2.5, _lib.dptr(x), _lib.dptr(x), _lib.dptr(x), None, 0, C.byref(bad),
                           None, None, None)
    assert rc == _lib.EINVAL and "struct_size" in _lib.last_error()
    ok = _lib.FitOptions()
    rc = lib.hbegp_fit_f64(None, _lib.dptr(x), _lib.dptr(x), 4, 1, 2.5, _lib.dptr(x), _lib.dptr(x), _lib.dptr(x), None, 0, C.byref(ok),
                           None, None, None)
    assert rc == _lib.EINVAL and "struct_size" not in _lib.last_error()  # the NULL context is what is wrong now


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert gpr.device_count() == 0
    with pytest.raises(_lib.HbegpError) as e:
        gpr.Context(1)
    assert e.value.code == _lib.ENODEV


def test_minimize_by_gradient_slanted_plane_kat():
    # gradmin.rs:75-101
    kat = KATS["slanted_plane"]
    f, x = gpr.minimize_by_gradient(lambda x: (x.sum(), np.ones_like(x)), kat["start"], kat["bounds"])
    assert x.tolist() == kat["x"] and f == kat["f"]


def test_minimize_by_gradient_bounded_quadratic_and_rosenbrock():
    # unconstrained optimum outside the box -> lands on the face; inside -> interior optimum
    c = np.array([3.0, -0.5, 0.25])
    f, x = gpr.minimize_by_gradient(lambda x: (((x - c) ** 2).sum(), 2 * (x - c)), [0.0, 0.0, 0.0], [(-1, 1)] * 3)
    np.testing.assert_allclose(x, [1.0, -0.5, 0.25], atol=1e-6)

    def rosen(x):
        f = 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2
        g = np.array([-400 * x[0] * (x[1] - x[0] ** 2) - 2 * (1 - x[0]), 200 * (x[1] - x[0] ** 2)])
        return f, g

    f, x = gpr.minimize_by_gradient(rosen, [-1.2, 1.0], [(-2, 2), (-2, 2)], maxeval=150)
    assert f < 1e-6 and np.allclose(x, [1, 1], atol=1e-3)


def test_minimize_respects_maxeval_and_infinite_values():
    calls = []

    def obj(x):
        calls.append(x.copy())
        if x[0] > 0.5:
            return np.inf, np.zeros_like(x)  # failed evaluation contract (fit.rs:105-112)
        return -(x[0]), -np.ones_like(x)

    f, x = gpr.minimize_by_gradient(obj, [0.0], [(-1, 1)], maxeval=40)
    assert len(calls) <= 40
    assert x[0] <= 0.5 and f <= -0.4
