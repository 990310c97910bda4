"""CPU restatement of hbetune's ``src/gpr`` Gaussian-process path.  TEST INFRASTRUCTURE ONLY.

This module is the parity *oracle*: a literal numpy/LAPACK transcription of the reference's
formulas.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; the product path (``hbetune_rs_amd`` / ``libhbegp.so``) never does.

Parity pinning: the reference is Rust (no toolchain in the build image) and its heavy arithmetic is
LAPACK ``?potrf/?potrs/?potri`` reached through ndarray-linalg 0.12.0 -> lapacke 0.2.0 ->
openblas-src 0.7.0 (Cargo.lock), none of which is vendored.  This restatement therefore calls the
same LAPACK routines through scipy's bundled OpenBLAS and is pinned by
  * the reference's own known-answer tables (matern_kernel.rs:200-214, :233-247,
    product_kernel.rs:137-163, cdist tests :285-305, clamp tests predict.rs:129-149), and
  * golden vectors generated with scikit-learn 1.7.2 (the library that produced the reference's
    constants) by ``tests/golden/make_golden.py`` (committed with the fixtures).
Cholesky/inverse digits and the optimiser trajectory are *unpinned by the reference itself*
(SURVEY.md 8c); see DESIGN.md.

Every function cites the reference file:line (relative to the reference root) it follows.
Element type ``A`` is float64 (default) or float32 (``--use-32``, main.rs:240-244); hyper-parameters
are always float64 and are converted with ``A::from_f`` (scalar.rs:3-30).
"""
import math

import numpy as np
from scipy.linalg import lapack

MIN_NOISE = 1e-5  # predict.rs:25 ``min_noise``


def _A(dtype):
    dt = np.dtype(dtype)
    assert dt in (np.dtype(np.float64), np.dtype(np.float32))
    return dt.type


# --------------------------------------------------------------------------------------------------
# bounded_value.rs:43-56  BoundedValue::with_clamped_value
def clamp(value, lo, hi):
    if value < lo:
        return lo
    if hi < value:
        return hi
    return value


# --------------------------------------------------------------------------------------------------
# matern_kernel.rs:262-283  cdist: r_ab = sqrt(sum_i (xa_ai - xb_bi)^2), accumulated in A
def cdist(xa, xb):
    assert xa.shape[1] == xb.shape[1]
    diff = xa[:, None, :] - xb[None, :, :]
    # sequential accumulation over the feature axis, like the reference's inner loop
    accum = np.zeros((xa.shape[0], xb.shape[0]), dtype=xa.dtype)
    for i in range(xa.shape[1]):
        accum += diff[:, :, i] ** 2
    return np.sqrt(accum)


# matern_kernel.rs:37-81  Matern::kernel  (nu in {0.5, 1.5, 2.5}; anything else is unimplemented!)
def matern_kernel(x1, x2, length_scale, nu):
    A = _A(x1.dtype)
    assert x1.shape[1] == len(length_scale) and x2.shape[1] == len(length_scale)
    ell = np.asarray(length_scale, dtype=np.float64).astype(A)  # :50 mapv(A::from_f)
    x1 = x1 / ell[None, :]  # :51-60
    x2 = x2 / ell[None, :]
    dists = cdist(x1, x2)  # :63
    if nu == 0.5:
        return np.exp(-dists)  # :66
    if nu == 1.5:
        k = dists * A(math.sqrt(3.0))  # :68
        return (k + A(1)) * np.exp(-k)  # :69
    if nu == 2.5:
        k = dists * A(math.sqrt(5.0))  # :73
        return (A(1) + k + k * k / A(3)) * np.exp(-k)  # :75
    if math.isinf(nu):
        # squared exponential (the nu -> inf limit).  NOT in the reference (:79 is unimplemented! for other nu): an extension
        # the north star names; its only oracle is sklearn's RBF (tests/golden/*_rbf.npz)
        return np.exp(-A(0.5) * dists * dists)
    raise NotImplementedError("Matern kernel with arbitrary values for nu")  # :79


# matern_kernel.rs:83-135  Matern::theta_grad -> (K [n,n], dK/dlog(ell) [n,n,d])
def matern_theta_grad(x, length_scale, nu):
    A = _A(x.dtype)
    kernel = matern_kernel(x, x, length_scale, nu)  # :84
    ell = np.asarray(length_scale, dtype=np.float64).astype(A)
    scales_sq = ell * ell  # :94
    d = (x[:, None, :] - x[None, :, :]) ** 2  # :95
    d = d / scales_sq[None, None, :]  # :96-98
    dsum = np.zeros(d.shape[:2], dtype=x.dtype)
    for k in range(d.shape[2]):  # sum_axis(Axis(2)), in feature order
        dsum += d[:, :, k]
    if nu == 0.5:
        with np.errstate(divide="ignore", invalid="ignore"):
            grad = kernel[:, :, None] * d / np.sqrt(dsum)[:, :, None]  # :103-104
        grad[~np.isfinite(grad)] = 0  # :105-109
    elif nu == 1.5:
        tmp = np.exp(-np.sqrt(dsum * A(3)))  # :114-116
        grad = d * tmp[:, :, None] * A(3)  # :117
    elif nu == 2.5:
        tmp = np.sqrt(dsum * A(5))[:, :, None]  # :120-122
        grad = np.exp(-tmp) * (tmp + A(1)) * d * A(5.0 / 3.0)  # :123-130
    elif math.isinf(nu):
        grad = np.exp(-A(0.5) * dsum)[:, :, None] * d  # dK/dlog(ell_k) = K * d_k  (extension, see matern_kernel)
    else:
        raise NotImplementedError("Matern kernel gradient with arbitrary values for nu")
    return kernel, grad


# constant_kernel.rs:24-42 + product_kernel.rs:36-38  K = c * K_matern
def product_kernel(x1, x2, amplitude, length_scale, nu):
    A = _A(x1.dtype)
    k1 = np.full((x1.shape[0], x2.shape[0]), A(amplitude), dtype=x1.dtype)  # constant_kernel.rs:24-29
    k2 = matern_kernel(x1, x2, length_scale, nu)
    return k1 * k2  # product_kernel.rs:37


# product_kernel.rs:40-70  theta_grad: grad[...,0] = dK1*K2 (= c*K2), grad[...,1+k] = dK2_k*K1
def product_theta_grad(x, amplitude, length_scale, nu):
    A = _A(x.dtype)
    n = x.shape[0]
    k1 = np.full((n, n), A(amplitude), dtype=x.dtype)
    g1 = np.full((n, n, 1), A(amplitude), dtype=x.dtype)  # constant_kernel.rs:31-38
    k2, g2 = matern_theta_grad(x, length_scale, nu)
    kernel = k1 * k2  # :56
    g1k2 = g1 * k2[:, :, None]  # :57
    g2k1 = g2 * k1[:, :, None]  # :58
    return kernel, np.concatenate([g1k2, g2k1], axis=2)  # :61-67


# product_kernel.rs:72-74  diag = c * 1
def product_diag(x, amplitude):
    return np.full(x.shape[0], _A(x.dtype)(amplitude), dtype=x.dtype)


# --------------------------------------------------------------------------------------------------
def _potrf(a):
    f = lapack.dpotrf if a.dtype == np.float64 else lapack.spotrf
    c, info = f(a, lower=1, clean=0, overwrite_a=0)
    return c, info


def _potrs(c, b):
    f = lapack.dpotrs if c.dtype == np.float64 else lapack.spotrs
    x, info = f(c, b, lower=1)
    assert info == 0
    return x


def _potri(c):
    f = lapack.dpotri if c.dtype == np.float64 else lapack.spotri
    inv, info = f(c, lower=1)
    assert info == 0
    # ndarray-linalg invc() returns the full symmetric matrix
    il = np.tril(inv)
    return il + np.tril(inv, -1).T


# lml.rs:29-79  lml_with_gradient.  Returns None when the Cholesky factorisation fails (:47-50).
# theta-gradient order: [noise, amplitude, ell_1..ell_d]  (:67-68)
def lml_with_gradient(x, y, noise, amplitude, length_scale, nu):
    A = _A(x.dtype)
    n = x.shape[0]
    kernel_matrix, kernel_gradient = product_theta_grad(x, amplitude, length_scale, nu)  # :40
    noise = A(noise)
    noise_gradient = np.eye(n, dtype=x.dtype) * noise  # :41
    kernel_matrix = kernel_matrix.copy()
    kernel_matrix[np.diag_indices(n)] += noise  # :44
    chol, info = _potrf(kernel_matrix)  # :47
    if info != 0:
        return None  # :48-50
    alpha = _potrs(chol, y)  # :54
    lml = (
        -0.5 * float(y.dot(alpha))
        - float(np.log(np.diag(chol)).sum())
        - n / 2.0 * math.log(2.0 * math.pi)
    )  # :57-59
    k_inv = _potri(chol)
    tmp = np.outer(alpha, alpha) - k_inv  # :62
    grads = [0.5 * float((tmp * noise_gradient).sum())]  # :67-70
    for j in range(kernel_gradient.shape[2]):
        grads.append(0.5 * float((tmp * kernel_gradient[:, :, j]).sum()))
    return dict(lml=lml, grad=np.array(grads, dtype=np.float64), alpha=alpha, chol=chol, k_inv=k_inv,
                kernel_matrix=kernel_matrix)


# fit.rs:93-134 objective contract, evaluated at a log-space theta = [ln s2, ln c, ln ell_1..]
# with bounds (linear space) bounds = [(lo,hi)] per parameter (kernel params are clamped :95, noise is not :96).
def objective(theta, x, y, nu, bounds):
    A = _A(x.dtype)
    noise = A(math.exp(theta[0]))
    c = clamp(math.exp(theta[1]), *bounds[1])
    ell = [clamp(math.exp(t), lo, hi) for t, (lo, hi) in zip(theta[2:], bounds[2:])]
    res = lml_with_gradient(x, y, noise, c, ell, nu)
    if res is None:
        return math.inf, np.zeros(len(theta)), None  # :105-112
    return -res["lml"], -res["grad"], res  # :128-133


# fit.rs:33-68  FittedKernel::extend : one evaluation at fixed parameters + invc_into
def extend(x, y, noise, amplitude, length_scale, nu):
    res = lml_with_gradient(x, y, noise, amplitude, length_scale, nu)
    if res is None:
        raise FloatingPointError("Kernel matrix must be invertible.")  # :55
    return res


# predict.rs:7-52
def predict(x, x_train, alpha, k_inv, amplitude, length_scale, nu, want_variance=True):
    A = _A(x_train.dtype)
    k_trans = product_kernel(x, x_train, amplitude, length_scale, nu)  # :18
    mean = k_trans.dot(alpha)  # :19
    if not want_variance:
        return mean, None, []
    min_noise = A(MIN_NOISE)
    y_var = product_diag(x, amplitude) + min_noise - np.einsum("ki,ki->k", k_trans.dot(k_inv), k_trans)  # :30-37
    below = clamp_negative_variance(y_var, -np.sqrt(min_noise))  # :39-48
    return mean, y_var, below


# predict.rs:104-127
def clamp_negative_variance(variances, warning_level):
    below = [v for v in variances if v < warning_level]
    variances[variances < 0] = 0
    return below


# --------------------------------------------------------------------------------------------------
# Adapter-level helpers used by the data generator (not on the device path).
# gpr.rs:429-450 estimate_amplitude -> (start, lo, hi)
def estimate_amplitude(y, bounds=None):
    y = np.asarray(y, dtype=np.float64)
    if bounds is None:
        hi = float((y ** 2).sum())
        ys = np.sort(y)
        # ndarray-stats Quantile1dExt::quantile_mut(0.1, Lower): index floor((n-1)*q)
        q = ys[int(math.floor((len(ys) - 1) * 0.1))]
        lo = q * q * len(y)
        lo = lo if lo > 2e-5 else 2e-5
        lo, hi = lo / 2.0, hi * 2.0
    else:
        lo, hi = bounds
    start = math.exp((math.log(lo) + math.log(hi)) / 2.0)
    return start, lo, hi


# ynormalize.rs:168-181 linear projection: (y - min) / mean(y - min) + 0.05   (see data generator)
def ynormalize_linear(y):
    y = np.asarray(y, dtype=np.float64)
    shifted = y - y.min()
    amp = shifted.mean()
    if amp <= 0:
        amp = 1.0
    return shifted / amp + 0.05
