"""ctypes loader for the plain-C oracle (oracle/gpr_oracle.c).  TEST INFRASTRUCTURE ONLY (see the C file's header)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libgpr_oracle.so")
_dp = C.POINTER(C.c_double)
_lib = None


def _nu2(nu):
    """2*nu as the C oracle's integer code; 0 = squared exponential (nu = infinity, an extension)."""
    import math

    return 0 if math.isinf(nu) else int(round(2 * nu))


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError(f"{_PATH} missing: run `make oracle` (or __graft_entry__.build())")
        lib = C.CDLL(_PATH)
        lib.oracle_lml_with_gradient.restype = C.c_int
        lib.oracle_lml_with_gradient.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _dp,
                                                 _dp, _dp, _dp, _dp, _dp, _dp]
        lib.oracle_predict.restype = C.c_int
        lib.oracle_predict.argtypes = [_dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp]
        lib.oracle_kernel.restype = None
        lib.oracle_kernel.argtypes = [_dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp]
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def lml_with_gradient(x, y, noise, amplitude, length_scale, nu):
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    ell = np.ascontiguousarray(length_scale, dtype=np.float64)
    n, d = x.shape
    lml = np.zeros(1)
    grad = np.zeros(d + 2)
    alpha = np.zeros(n)
    kinv = np.zeros((n, n))
    kmat = np.zeros((n, n))
    ldiag = np.zeros(n)
    st = lib.oracle_lml_with_gradient(_p(x), _p(y), n, d, _nu2(nu), float(noise), float(amplitude), _p(ell),
                                      _p(lml), _p(grad), _p(alpha), _p(kinv), _p(kmat), _p(ldiag))
    if st != 0:
        return None
    return dict(lml=float(lml[0]), grad=grad, alpha=alpha, k_inv=kinv, kernel_matrix=kmat, ldiag=ldiag)


def predict(xs, x_train, alpha, k_inv, amplitude, length_scale, nu, want_variance=True):
    lib = load()
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    x_train = np.ascontiguousarray(x_train, dtype=np.float64)
    ell = np.ascontiguousarray(length_scale, dtype=np.float64)
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    k_inv = np.ascontiguousarray(k_inv, dtype=np.float64)
    m, d = xs.shape
    mean = np.zeros(m)
    var = np.zeros(m) if want_variance else None
    warn = lib.oracle_predict(_p(xs), m, _p(x_train), x_train.shape[0], d, _nu2(nu), float(amplitude), _p(ell),
                              _p(alpha), _p(k_inv), _p(mean), _p(var))
    return mean, var, warn
