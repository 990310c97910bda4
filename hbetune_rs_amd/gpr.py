"""Host-side mirror of the reference's `src/gpr` interface over the C ABI (include/hbegp.h).

Names follow the reference: `FittedKernel.new / .extend` (src/gpr/fit.rs:18-68), `predict` (src/gpr/predict.rs:7-52),
`LmlWithGradient.of` (src/gpr/lml.rs:16-27).  All arithmetic happens in libhbegp.so on the GPU; this file only
marshals arrays.  Python is used here because the reference's own toolchain (Rust) is absent from the build image;
the Rust binding a maintainer would write is in INTEGRATION.md.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import HbegpError, NOT_PD, ALL_FAILED  # noqa: F401


def _suffix(dtype):
    dt = np.dtype(dtype)
    if dt == np.float64:
        return "f64"
    if dt == np.float32:
        return "f32"
    raise TypeError(f"element type must be float64 or float32, got {dt}")


class Context:
    """Owns the GPUs used by fits (hbegp_ctx)."""

    def __init__(self, n_devices=1, device_ids=None):
        lib = _lib.load()
        self._h = C.c_void_p()
        ids = None
        if device_ids is not None:
            ids = (C.c_int * len(device_ids))(*device_ids)
            n_devices = len(device_ids)
        _lib.check(lib.hbegp_ctx_create(n_devices, ids, C.byref(self._h)))
        self.n_devices = n_devices

    def close(self):
        if self._h:
            _lib.load().hbegp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(1)
    return _default_ctx


def device_count():
    return _lib.load().hbegp_device_count()


class Problem:
    """X, y resident on the device; repeated lml/gradient evaluations (the optimiser's inner loop)."""

    def __init__(self, x_train, y_train, nu=2.5, n_slots=1, ctx=None):
        lib = _lib.load()
        self.ctx = ctx or default_context()
        self.dtype = np.dtype(x_train.dtype)
        sfx = _suffix(self.dtype)
        self.x = _lib.as_c(x_train, self.dtype)
        self.y = _lib.as_c(y_train, self.dtype)
        assert self.x.ndim == 2 and self.y.shape == (self.x.shape[0],)
        self.n, self.d = self.x.shape
        self.p = self.d + 2
        self._sfx = sfx
        self._h = C.c_void_p()
        create = getattr(lib, f"hbegp_problem_create_{sfx}")
        _lib.check(create(self.ctx._h, _lib.aptr(self.x), _lib.aptr(self.y), self.n, self.d, float(nu), n_slots, C.byref(self._h)))

    def close(self):
        if self._h:
            _lib.load().hbegp_problem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lml_with_gradient(self, theta, lo=None, hi=None, want_grad=True, dev=0, slot=0):
        """LmlWithGradient::of (lml.rs:16-79) at log-space theta.  Returns (lml, grad) or None when not PD."""
        lib = _lib.load()
        theta = _lib.as_c(theta, np.float64)
        lo = None if lo is None else _lib.as_c(lo, np.float64)
        hi = None if hi is None else _lib.as_c(hi, np.float64)
        lml = C.c_double()
        grad = np.zeros(self.p) if want_grad else None
        code = lib.hbegp_problem_eval(self._h, dev, slot, _lib.dptr(theta), _lib.dptr(lo), _lib.dptr(hi), C.byref(lml), _lib.dptr(grad))
        if _lib.check(code, allow=(NOT_PD,)) == NOT_PD:
            return None
        return lml.value, grad

    def results(self, want_kinv=True, dev=0, slot=0):
        """(alpha, k_inv, diag(L)) of the most recent evaluation."""
        lib = _lib.load()
        alpha = np.zeros(self.n, dtype=self.dtype)
        ldiag = np.zeros(self.n, dtype=self.dtype)
        kinv = np.zeros((self.n, self.n), dtype=self.dtype) if want_kinv else None
        get = getattr(lib, f"hbegp_problem_get_{self._sfx}")
        _lib.check(get(self._h, dev, slot, _lib.aptr(alpha), _lib.aptr(kinv), _lib.aptr(ldiag)))
        return alpha, kinv, ldiag

    def debug_work_matrix(self, which, dev=0, slot=0):
        """Raw copy of W1 (which=1) or W2 (which=2) after the last evaluation: [np, np] with np = n rounded up to 128."""
        npad = (self.n + 127) // 128 * 128
        out = np.zeros((npad, npad), dtype=self.dtype)
        get = getattr(_lib.load(), f"hbegp_problem_debug_get_{self._sfx}")
        _lib.check(get(self._h, dev, slot, which, _lib.aptr(out)))
        return out

    def kernel_matrix(self, theta, lo=None, hi=None, dev=0, slot=0):
        lib = _lib.load()
        theta = _lib.as_c(theta, np.float64)
        lo = None if lo is None else _lib.as_c(lo, np.float64)
        hi = None if hi is None else _lib.as_c(hi, np.float64)
        K = np.zeros((self.n, self.n), dtype=self.dtype)
        kmat = getattr(lib, f"hbegp_problem_kmat_{self._sfx}")
        _lib.check(kmat(self._h, dev, slot, _lib.dptr(theta), _lib.dptr(lo), _lib.dptr(hi), _lib.aptr(K)))
        return K

    def time_eval(self, theta, reps=5, dev=0, slot=0):
        lib = _lib.load()
        theta = _lib.as_c(theta, np.float64)
        phases = np.zeros(24)
        _lib.check(lib.hbegp_problem_time_eval(self._h, dev, slot, _lib.dptr(theta), reps, _lib.dptr(phases)))
        keys = ["kmat_ms", "chol_gemm_ms", "leaf_ms", "lauum_ms", "alpha_ms", "gradtrace_ms", "eval_graph_ms", "n_gemm",
                "gemm128_ms", "gemm128_gflop", "gemm64_ms", "gemm64_gflop", "gemm32_ms", "gemm32_gflop", "eval_eager_ms", "n_leaf",
                "n_gemm128", "n_gemm64", "n_gemm32", "dag_ms", "dag_gflop", "r1", "r2", "r3"]
        return dict(zip(keys, phases.tolist()))

    def time_concurrent(self, theta, reps=5, dev=0):
        """Every slot evaluating at once (the configuration a fit runs in): wall time per round and per-kernel-class times
        from hipEvents on each slot's stream."""
        lib = _lib.load()
        theta = _lib.as_c(theta, np.float64)
        out = np.zeros(16)
        _lib.check(lib.hbegp_problem_time_concurrent(self._h, dev, _lib.dptr(theta), reps, _lib.dptr(out)))
        keys = ["round_ms", "n_slots", "round_eager_ms", "kmat_ms", "factor_ms", "factor_gflop", "lauum_ms", "lauum_gflop", "alpha_ms",
                "gradtrace_ms", "task_queue_workgroups", "factor_launches"]
        return dict(zip(keys, out.tolist()))


class FittedKernel:
    """Mirror of `FittedKernel<K, A>` (fit.rs:6-12): kernel parameters, noise, alpha, k_inv, lml + the model handle."""

    def __init__(self, handle, dtype, n, d, nu):
        self._h = handle
        self.dtype = np.dtype(dtype)
        self.n, self.d, self.nu = n, d, nu
        self._sfx = _suffix(dtype)
        lml = C.c_double()
        _lib.check(_lib.load().hbegp_model_info(self._h, None, None, None, None, C.byref(lml)))
        self.lml = lml.value
        self.theta = np.zeros(d + 2)
        get = getattr(_lib.load(), f"hbegp_model_get_{self._sfx}")
        _lib.check(get(self._h, _lib.dptr(self.theta), None, None))

    # -- parameters in the reference's terms --
    @property
    def noise(self):
        return math.exp(self.theta[0])

    @property
    def amplitude(self):
        return math.exp(self.theta[1])

    @property
    def length_scale(self):
        return np.exp(self.theta[2:])

    def arrays(self, want_kinv=True):
        """(alpha, k_inv) copied from the device; k_inv is the full symmetric matrix (fit.rs:60,168)."""
        alpha = np.zeros(self.n, dtype=self.dtype)
        kinv = np.zeros((self.n, self.n), dtype=self.dtype) if want_kinv else None
        get = getattr(_lib.load(), f"hbegp_model_get_{self._sfx}")
        _lib.check(get(self._h, None, _lib.aptr(alpha), _lib.aptr(kinv)))
        return alpha, kinv

    @staticmethod
    def new(x_train, y_train, theta0, lo, hi, starts=None, nu=2.5, ctx=None, maxeval=150, fixed_work=False, trace=False):
        """FittedKernel::new (fit.rs:18-31, 71-176): 1 + len(starts) bounded L-BFGS runs, capture the best lml."""
        lib = _lib.load()
        ctx = ctx or default_context()
        dtype = np.dtype(x_train.dtype)
        sfx = _suffix(dtype)
        x = _lib.as_c(x_train, dtype)
        y = _lib.as_c(y_train, dtype)
        n, d = x.shape
        p = d + 2
        theta0 = _lib.as_c(theta0, np.float64)
        lo = _lib.as_c(lo, np.float64)
        hi = _lib.as_c(hi, np.float64)
        assert theta0.shape == (p,) and lo.shape == (p,) and hi.shape == (p,)
        n_restarts = 0 if starts is None else len(starts)
        starts_c = None if n_restarts == 0 else _lib.as_c(np.asarray(starts).reshape(n_restarts, p), np.float64)
        opt = _lib.FitOptions()
        opt.maxeval = maxeval
        opt.fixed_work = 1 if fixed_work else 0
        n_evals, n_not_pd = C.c_int(0), C.c_int(0)
        opt.n_evals = C.pointer(n_evals)
        opt.n_not_pd = C.pointer(n_not_pd)
        tr = None
        if trace:
            cap = (1 + n_restarts) * maxeval
            tr = dict(theta=np.zeros((cap, p)), lml=np.zeros(cap), grad=np.zeros((cap, p)), run=np.zeros(cap, dtype=np.int32),
                      count=C.c_int(0))
            opt.trace_cap = cap
            opt.trace_theta = _lib.dptr(tr["theta"])
            opt.trace_lml = _lib.dptr(tr["lml"])
            opt.trace_grad = _lib.dptr(tr["grad"])
            opt.trace_run = tr["run"].ctypes.data_as(C.POINTER(C.c_int))
            opt.trace_count = C.pointer(tr["count"])
        handle = C.c_void_p()
        theta_best = np.zeros(p)
        lml_best = C.c_double()
        fit = getattr(lib, f"hbegp_fit_{sfx}")
        _lib.check(fit(ctx._h, _lib.aptr(x), _lib.aptr(y), n, d, float(nu), _lib.dptr(theta0), _lib.dptr(lo), _lib.dptr(hi),
                       _lib.dptr(starts_c), n_restarts, C.byref(opt), _lib.dptr(theta_best), C.byref(lml_best), C.byref(handle)))
        fk = FittedKernel(handle, dtype, n, d, nu)
        fk.n_evals, fk.n_not_pd = n_evals.value, n_not_pd.value
        if trace:
            k = tr["count"].value
            fk.trace = dict(theta=tr["theta"][:k], lml=tr["lml"][:k], grad=tr["grad"][:k], run=tr["run"][:k])
        return fk

    @staticmethod
    def extend(x_train, y_train, theta, lo=None, hi=None, nu=2.5, ctx=None):
        """FittedKernel::extend (fit.rs:33-68): one evaluation at fixed theta + K^-1.  Raises where the reference panics."""
        lib = _lib.load()
        ctx = ctx or default_context()
        dtype = np.dtype(x_train.dtype)
        sfx = _suffix(dtype)
        x = _lib.as_c(x_train, dtype)
        y = _lib.as_c(y_train, dtype)
        n, d = x.shape
        theta = _lib.as_c(theta, np.float64)
        lo = None if lo is None else _lib.as_c(lo, np.float64)
        hi = None if hi is None else _lib.as_c(hi, np.float64)
        handle = C.c_void_p()
        ext = getattr(lib, f"hbegp_extend_{sfx}")
        _lib.check(ext(ctx._h, _lib.aptr(x), _lib.aptr(y), n, d, float(nu), _lib.dptr(theta), _lib.dptr(lo), _lib.dptr(hi),
                       C.byref(handle)))
        return FittedKernel(handle, dtype, n, d, nu)

    def extend_with(self, x_train, y_train, ctx=None):
        """FittedKernel::extend (fit.rs:33-68) at this model's theta on data whose leading rows are this model's training
        rows (minimize.rs:629-644 appends the validation samples): reuses the factorisation of the kept 128-blocks, O(n^2 k)
        instead of O(n^3).  Falls back to the full path when the prefix differs.  Sets `.incremental` on the result."""
        lib = _lib.load()
        ctx = ctx or default_context()
        x = _lib.as_c(x_train, self.dtype)
        y = _lib.as_c(y_train, self.dtype)
        n, d = x.shape
        assert d == self.d
        handle = C.c_void_p()
        inc = C.c_int(0)
        ext = getattr(lib, f"hbegp_extend_from_{self._sfx}")
        _lib.check(ext(ctx._h, self._h, _lib.aptr(x), _lib.aptr(y), n, C.byref(handle), C.byref(inc)))
        fk = FittedKernel(handle, self.dtype, n, d, self.nu)
        fk.incremental = bool(inc.value)
        return fk

    def predict(self, x, want_variance=True):
        """predict() (predict.rs:7-52): returns (mean, variance or None, n_warn)."""
        lib = _lib.load()
        x = _lib.as_c(x, self.dtype)
        assert x.ndim == 2 and x.shape[1] == self.d
        m = x.shape[0]
        mean = np.zeros(m, dtype=self.dtype)
        var = np.zeros(m, dtype=self.dtype) if want_variance else None
        n_warn = C.c_int(0)
        pred = getattr(lib, f"hbegp_predict_{self._sfx}")
        _lib.check(pred(self._h, _lib.aptr(x), m, _lib.aptr(mean), _lib.aptr(var), C.byref(n_warn)))
        return mean, var, n_warn.value

    def release(self):
        if self._h:
            _lib.load().hbegp_model_release(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def minimize_by_gradient(objective, x0, bounds, maxeval=150):
    """util::minimize_by_gradient (gradmin.rs:35-60) through the library's bounded L-BFGS."""
    lib = _lib.load()
    x = _lib.as_c(np.array(x0, dtype=np.float64), np.float64).copy()
    n = len(x)
    lo = _lib.as_c([b[0] for b in bounds], np.float64)
    hi = _lib.as_c([b[1] for b in bounds], np.float64)

    def cb(xp, gp, _user):
        xs = np.ctypeslib.as_array(xp, shape=(n,))
        f, g = objective(xs.copy())
        gv = np.ctypeslib.as_array(gp, shape=(n,))
        gv[:] = g
        return float(f)

    fn = _lib.OBJECTIVE_FN(cb)
    f = lib.hbegp_minimize_by_gradient(fn, None, _lib.dptr(x), _lib.dptr(lo), _lib.dptr(hi), n, maxeval)
    return f, x
