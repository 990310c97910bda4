"""ctypes binding of libhbegp.so (the C ABI in include/hbegp.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make``.  There is no CPU fallback: if the shared
object is missing, or no gfx950 device is present, the calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HBEGP_LIB") or os.path.join(HERE, "libhbegp.so")  # HBEGP_LIB: an experimental build of the same library

OK, NOT_PD, ALL_FAILED = 0, 1, 2
EINVAL, EHIP, ENODEV, ENOMEM = -1, -2, -3, -4


class HbegpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hbegp error {code}: {msg}")
        self.code = code


class FitOptions(C.Structure):
    """hbegp_fit_options; `struct_size` is filled in by __init__ (the library copies min(struct_size, its own size))."""

    _fields_ = [
        ("struct_size", C.c_size_t),
        ("maxeval", C.c_int),
        ("fixed_work", C.c_int),
        ("lbfgs_memory", C.c_int),
        ("trace_cap", C.c_int),
        ("trace_theta", C.POINTER(C.c_double)),
        ("trace_lml", C.POINTER(C.c_double)),
        ("trace_grad", C.POINTER(C.c_double)),
        ("trace_run", C.POINTER(C.c_int)),
        ("trace_count", C.POINTER(C.c_int)),
        ("n_evals", C.POINTER(C.c_int)),
        ("n_not_pd", C.POINTER(C.c_int)),
    ]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(FitOptions)


OBJECTIVE_FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# symbol -> (restype, argtypes); every symbol include/hbegp.h declares
SIGNATURES = {
    "hbegp_version": (C.c_int, []),
    "hbegp_device_count": (C.c_int, []),
    "hbegp_ctx_create": (C.c_int, [C.c_int, _ip, C.POINTER(_vp)]),
    "hbegp_ctx_destroy": (None, [_vp]),
    "hbegp_last_error": (C.c_char_p, []),
    "hbegp_problem_create_f64": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(_vp)]),
    "hbegp_problem_create_f32": (C.c_int, [_vp, _fp, _fp, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(_vp)]),
    "hbegp_problem_destroy": (None, [_vp]),
    "hbegp_problem_eval": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "hbegp_problem_get_f64": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp]),
    "hbegp_problem_get_f32": (C.c_int, [_vp, C.c_int, C.c_int, _fp, _fp, _fp]),
    "hbegp_problem_debug_get_f64": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp]),
    "hbegp_problem_debug_get_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _fp]),
    "hbegp_problem_kmat_f64": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp]),
    "hbegp_problem_kmat_f32": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _fp]),
    "hbegp_problem_time_eval": (C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_int, _dp]),
    "hbegp_problem_time_concurrent": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp]),
    "hbegp_fit_f64": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp, C.c_int,
                                C.POINTER(FitOptions), _dp, _dp, C.POINTER(_vp)]),
    "hbegp_fit_f32": (C.c_int, [_vp, _fp, _fp, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp, C.c_int,
                                C.POINTER(FitOptions), _dp, _dp, C.POINTER(_vp)]),
    "hbegp_extend_f64": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, C.POINTER(_vp)]),
    "hbegp_extend_f32": (C.c_int, [_vp, _fp, _fp, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, C.POINTER(_vp)]),
    "hbegp_extend_from_f64": (C.c_int, [_vp, _vp, _dp, _dp, C.c_int, C.POINTER(_vp), _ip]),
    "hbegp_extend_from_f32": (C.c_int, [_vp, _vp, _fp, _fp, C.c_int, C.POINTER(_vp), _ip]),
    "hbegp_predict_f64": (C.c_int, [_vp, _dp, C.c_int, _dp, _dp, _ip]),
    "hbegp_predict_f32": (C.c_int, [_vp, _fp, C.c_int, _fp, _fp, _ip]),
    "hbegp_model_info": (C.c_int, [_vp, _ip, _ip, _ip, _dp, _dp]),
    "hbegp_model_get_f64": (C.c_int, [_vp, _dp, _dp, _dp]),
    "hbegp_model_get_f32": (C.c_int, [_vp, _dp, _fp, _fp]),
    "hbegp_model_retain": (None, [_vp]),
    "hbegp_model_release": (None, [_vp]),
    "hbegp_debug_lbfgs_replay": (C.c_int, [C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _ip]),
    "hbegp_debug_dag_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _dp, _dp, _dp, C.c_char_p,
                                       C.c_int]),
    "hbegp_minimize_by_gradient": (C.c_double, [OBJECTIVE_FN, _vp, _dp, _dp, _dp, C.c_int, C.c_int]),
}

_lib = None


def load():
    """Load libhbegp.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make`. "
            "hbetune_rs_amd has no CPU fallback."
        )
    # PyTorch bundles its own HIP runtime (same soname).  If torch is going to live in this process, it has to be
    # loaded first so that there is exactly one libamdhip64 in the process.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().hbegp_last_error().decode("utf-8", "replace")


def check(code, allow=()):
    if code != OK and code not in allow:
        raise HbegpError(code, last_error())
    return code


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def fptr(a):
    return None if a is None else a.ctypes.data_as(_fp)


def aptr(a):
    """Pointer of matching element type for an f32/f64 array."""
    if a is None:
        return None
    return fptr(a) if a.dtype == np.float32 else dptr(a)


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
