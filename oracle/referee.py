"""Extended-precision referee for ill-conditioned kernel matrices.  TEST INFRASTRUCTURE ONLY (see oracle/referee.c).

Where a fit ends, cond(K) reaches 1e11..1e12 and the GPU engine and the LAPACK oracle differ by ~cond(K) * eps; neither is the
truth there.  The referee computes alpha = K^-1 y (lml.rs:54) and the predictive mean / variance (predict.rs:18-37) to ~1e-18:
K and k* are evaluated in IEEE binary128 from the f64 inputs (taken as exact) and every linear system is solved by
mixed-precision iterative refinement -- the f64 LAPACK factor of K as preconditioner, residuals in double-double arithmetic, the
iterate carried as a double-double vector.  The refinement contracts by ~cond(K) * eps_f64 per sweep, so it needs
cond(K) well below 1e15 (it raises otherwise).

Only tests/ and bench.py's parity leg import this; the product never does.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np
from scipy.linalg import lapack

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libreferee.so")
MIN_NOISE = 1e-5  # predict.rs:25

_dp = C.POINTER(C.c_double)


def build():
    src = os.path.join(HERE, "referee.c")
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return LIB
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-mfma", "-ffp-contract=off", "-o", LIB, src, "-lquadmath", "-lm"])
    return LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        lib.referee_create.restype = C.c_void_p
        lib.referee_create.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double, _dp, C.c_int]
        lib.referee_destroy.argtypes = [C.c_void_p]
        lib.referee_khi.argtypes = [C.c_void_p, _dp]
        lib.referee_residual.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]
        lib.referee_kstar.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp]
        lib.referee_coldot.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        lib.referee_axpy.argtypes = [C.c_int, _dp, _dp, _dp]
        lib.referee_logdet.restype = C.c_int
        lib.referee_logdet.argtypes = [C.c_void_p, _dp]
        lib.referee_gradient.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp]
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def nu2_of(nu):
    return 0 if math.isinf(nu) else int(round(2 * nu))


class Referee:
    """Truth for one (X, y, noise, amplitude, length_scale, nu): alpha, predictive mean / variance, selected columns of K^-1."""

    def __init__(self, X, y, noise, amplitude, length_scale, nu=2.5):
        lib = _load()
        self.X = _c(X)
        self.y = _c(y)
        self.n, self.d = self.X.shape
        self.amp, self.noise = float(amplitude), float(noise)
        self.ell = _c(length_scale)
        self.h = lib.referee_create(_p(self.X), self.n, self.d, self.noise, self.amp, _p(self.ell), nu2_of(nu))
        if not self.h:
            raise MemoryError("referee_create")
        khi = np.empty((self.n, self.n))
        lib.referee_khi(self.h, _p(khi))
        self.khi = khi
        self.chol, info = lapack.dpotrf(khi, lower=1, clean=0, overwrite_a=0)
        if info != 0:
            raise FloatingPointError("the f64 preconditioner is not positive definite: cond(K) is beyond the referee's reach")
        self.sweeps = []  # per solve: the relative size of every correction

    def close(self):
        if self.h:
            _load().referee_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, bhi, blo=None, maxit=60):
        """K x = b for b = bhi + blo ([n] or [n, m]); returns (xhi, xlo) with x = xhi + xlo good to ~cond(K) * 1e-30."""
        lib = _load()
        bhi = _c(bhi)
        one = bhi.ndim == 1
        B = bhi.reshape(self.n, -1).copy()
        Bl = None if blo is None else _c(blo).reshape(self.n, -1).copy()
        m = B.shape[1]
        xh, xl = np.zeros_like(B), np.zeros_like(B)
        r = np.empty_like(B)
        hist = []
        prev = math.inf
        for _ in range(maxit):
            lib.referee_residual(self.h, m, _p(B), None if Bl is None else _p(Bl), _p(xh), _p(xl), _p(r))
            dlt, info = lapack.dpotrs(self.chol, r, lower=1)
            assert info == 0
            dlt = _c(dlt)
            lib.referee_axpy(dlt.size, _p(xh), _p(xl), _p(dlt))
            size = float(np.max(np.abs(dlt)) / max(np.max(np.abs(xh)), 1e-300))
            hist.append(size)
            if size < 1e-29 or (size >= prev and size < 1e-20):  # converged, or at the floor of the double-double residual
                break
            if len(hist) >= 4 and size > 0.5 * prev and size > 1e-12:
                raise FloatingPointError(f"iterative refinement does not contract (corrections {hist}): cond(K) is beyond the referee's reach")
            prev = size
        else:
            raise FloatingPointError(f"iterative refinement did not converge in {maxit} sweeps: {hist}")
        self.sweeps.append(hist)
        return (xh[:, 0], xl[:, 0]) if one else (xh, xl)

    def alpha(self):
        """alpha = K^-1 y (lml.rs:54) as a double-double pair."""
        if not hasattr(self, "_alpha"):
            self._alpha = self.solve(self.y)
        return self._alpha

    def kinv(self):
        """The full K^-1 (fit.rs:168 invc_into) as a double-double pair of [n, n] arrays (n solves: for test sizes)."""
        if not hasattr(self, "_kinv"):
            self._kinv = self.solve(np.eye(self.n))
        return self._kinv

    def lml(self):
        """lml.rs:57-59: -1/2 y^T alpha - sum log L_ii - n/2 log(2 pi), the log-determinant from a double-double Cholesky."""
        lib = _load()
        ah, al = self.alpha()
        yh, yl = np.empty(1), np.empty(1)
        z = np.zeros(self.n)
        lib.referee_coldot(self.n, 1, _p(self.y), _p(z), _p(_c(ah)), _p(_c(al)), _p(yh), _p(yl))
        ld_ = np.empty(2)
        info = lib.referee_logdet(self.h, _p(ld_))
        if info != 0:
            raise FloatingPointError(f"double-double Cholesky failed at pivot {info - 1}")
        ld = np.longdouble
        v = -(ld(yh[0]) + ld(yl[0])) / 2 - (ld(ld_[0]) + ld(ld_[1])) / 2 - ld(self.n) / 2 * np.log(2 * ld(np.pi))
        return float(v)

    def gradient(self):
        """lml.rs:62-70 in binary128 with the refined alpha and K^-1: [noise, amplitude, ell_1..ell_d]."""
        lib = _load()
        ah, al = self.alpha()
        vh, vl = self.kinv()
        p = self.d + 2
        gh, gl = np.empty(p), np.empty(p)
        lib.referee_gradient(self.h, _p(_c(ah)), _p(_c(al)), _p(_c(vh)), _p(_c(vl)), _p(gh), _p(gl))
        return gh + gl

    def kinv_columns(self, cols):
        """Columns `cols` of K^-1 (fit.rs:168 invc_into), rounded to f64: [n, len(cols)]."""
        E = np.zeros((self.n, len(cols)))
        for q, j in enumerate(cols):
            E[j, q] = 1.0
        xh, xl = self.solve(E)
        return xh + xl

    def predict(self, Xs):
        """predict.rs:7-52 in extended precision: (mean, variance clamped at 0 as :39-48, raw variance), each rounded to f64."""
        lib = _load()
        Xs = _c(Xs)
        m = Xs.shape[0]
        kh, kl = np.empty((self.n, m)), np.empty((self.n, m))
        lib.referee_kstar(self.h, _p(Xs), m, _p(kh), _p(kl))
        ah, al = self.alpha()
        Ah, Al = _c(np.repeat(ah[:, None], m, axis=1)), _c(np.repeat(al[:, None], m, axis=1))
        mh, ml = np.empty(m), np.empty(m)
        lib.referee_coldot(self.n, m, _p(kh), _p(kl), _p(Ah), _p(Al), _p(mh), _p(ml))
        mean = mh + ml
        wh, wl = self.solve(kh, kl)
        qh, ql = np.empty(m), np.empty(m)
        lib.referee_coldot(self.n, m, _p(kh), _p(kl), _p(_c(wh)), _p(_c(wl)), _p(qh), _p(ql))
        ld = np.longdouble
        raw = ((ld(self.amp) + ld(MIN_NOISE)) - ld(qh)) - ld(ql)  # diag + min_noise - k*^T K^-1 k*  (predict.rs:30-37)
        raw = raw.astype(np.float64)
        return mean, np.where(raw < 0, 0.0, raw), raw
