"""Deterministic synthetic workloads for the GP path (bench + parity tests).

Mirrors what the reference's caller hands to ``Estimator::estimate``:
  * X: n x d row-major features in [0,1]^d (``Space::project_into_features_array``, space.rs:141-159),
  * y: a benchmark function of the natural-units point (benchfn.rs) followed by the linear or
    logarithmic y-normalisation of ynormalize.rs:162-195.
The generator is SplitMix64 (state = seed, standard constants), 53-bit mantissa -> [0,1).
Nothing here touches the GPU or the oracle.
"""
import math

import numpy as np

MASK = (1 << 64) - 1


def splitmix64_uniform(seed, count):
    """`count` doubles in [0,1) from SplitMix64 started at `seed`."""
    out = np.empty(count, dtype=np.float64)
    state = seed & MASK
    for i in range(count):
        state = (state + 0x9E3779B97F4A7C15) & MASK
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        z = z ^ (z >> 31)
        out[i] = (z >> 11) * (1.0 / (1 << 53))
    return out


def splitmix64_uniform_fast(seed, count):
    """Vectorised SplitMix64 (identical stream to `splitmix64_uniform`)."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


# ---- benchfn.rs restated (y generators only) ------------------------------------------------------
def sphere(x):  # benchfn.rs:19-26
    return (x ** 2).sum(axis=1)


def goldstein_price(x):  # benchfn.rs:42-58
    x1, x2 = x[:, 0], x[:, 1]
    a = 1 + (x1 + x2 + 1) ** 2 * (19 - 14 * x1 + 3 * x1 ** 2 - 14 * x2 + 6 * x1 * x2 + 3 * x2 ** 2)
    b = 30 + (2 * x1 - 3 * x2) ** 2 * (18 - 32 * x1 + 12 * x1 ** 2 + 48 * x2 - 36 * x1 * x2 + 27 * x2 ** 2)
    return a * b


def himmelblau(x):  # benchfn.rs:102-108
    x1, x2 = x[:, 0], x[:, 1]
    return (x1 ** 2 + x2 - 11) ** 2 + (x1 + x2 ** 2 - 7) ** 2


def rastrigin(x, amplitude=10.0):  # benchfn.rs:133-146
    return amplitude * x.shape[1] + (x ** 2 - amplitude * np.cos(2 * math.pi * x)).sum(axis=1)


def rosenbrock(x):  # benchfn.rs:169-191
    return (100.0 * (x[:, 1:] - x[:, :-1] ** 2) ** 2 + (1 - x[:, :-1]) ** 2).sum(axis=1)


BENCHFNS = {
    # name: (function, lo, hi)
    "sphere": (sphere, -2.0, 2.0),
    "goldstein_price": (goldstein_price, -2.0, 2.0),
    "himmelblau": (himmelblau, -5.0, 5.0),
    "rastrigin": (rastrigin, -5.12, 5.12),
    "rosenbrock": (rosenbrock, -2.5, 2.5),
}


# ---- ynormalize.rs:162-195 ------------------------------------------------------------------------
def ynormalize(y, projection="linear"):
    y = np.asarray(y, dtype=np.float64)
    if projection == "linear":
        expected = y.min()  # guess_min(.., minimum = 0)  ynormalize.rs:291-303
        y = y - expected
        amp = y.mean()  # guess_amplitude ynormalize.rs:307-322
        amp = amp if amp > 0 else 1.0
        return y / amp + 0.05  # FUDGE_MIN
    if projection == "logarithmic":
        expected = y.min() - 1.0
        y = np.log(y - expected)
        amp = y.mean()
        amp = amp if amp > 0 else 1.0
        return y / amp
    raise ValueError(projection)


# gpr.rs:429-450 estimate_amplitude -> (start, lo, hi)
def estimate_amplitude(y):
    y = np.asarray(y, dtype=np.float64)
    hi = float((y ** 2).sum())
    ys = np.sort(y)
    q = ys[int(math.floor((len(ys) - 1) * 0.1))]  # Quantile1dExt::quantile_mut(0.1, Lower)
    lo = q * q * len(y)
    lo = lo if lo > 2e-5 else 2e-5
    lo, hi = lo / 2.0, hi * 2.0
    return math.exp((math.log(lo) + math.log(hi)) / 2.0), lo, hi


# ---- BASELINE.json configs --------------------------------------------------------------------------
CONFIGS = {
    #        fn                d   n     dtype      projection     seed index
    "C1": ("sphere", 2, 64, "float64", "linear", 1),
    "C2": ("rosenbrock", 8, 1024, "float64", "linear", 2),
    "C3": ("rastrigin", 16, 4096, "float64", "linear", 3),
    "C4": ("goldstein_price", 2, 8192, "float64", "logarithmic", 4),
    "C5": ("himmelblau", 2, 2048, "float32", "linear", 5),
    "M": ("rosenbrock", 8, 4096, "float64", "linear", 6),
}
SEED_BASE = 0xC0FFEE


def make_workload(name, n=None, dtype=None):
    """Return dict(X, y, nu, theta, bounds_lo, bounds_hi, ...) for a BASELINE config (optionally at reduced n).

    theta is the fixed, well-conditioned parity/throughput point of SURVEY.md 8(d):
    sigma^2 = 1e-2*c, c = estimate_amplitude start value, ell_k = 0.3 + 0.1*k/d;
    log-space order [ln s2, ln c, ln ell_1..ell_d] (fit.rs:140-144).
    """
    fn_name, d, n_full, dt, projection, idx = CONFIGS[name]
    n = n_full if n is None else n
    dt = np.dtype(dt if dtype is None else dtype)
    u = splitmix64_uniform_fast(SEED_BASE + idx, n * d).reshape(n, d)
    fn, lo, hi = BENCHFNS[fn_name]
    y_nat = fn(lo + (hi - lo) * u)
    y = ynormalize(y_nat, projection)
    c0, c_lo, c_hi = estimate_amplitude(y)
    ell = np.array([0.3 + 0.1 * k / d for k in range(d)])
    theta = np.concatenate([[math.log(1e-2 * c0), math.log(c0)], np.log(ell)])
    # default bounds of EstimatorGPR::new gpr.rs:219-236 (linear space)
    lo_b = np.concatenate([[1e-5, c_lo], np.full(d, 1e-3)])
    hi_b = np.concatenate([[1e5, c_hi], np.full(d, 1e3)])
    # default start of get_kernel_or_default gpr.rs:402-427: noise 1, ell = geometric mean of bounds (=1)
    theta0 = np.concatenate([[0.0, math.log(c0)], np.zeros(d)])
    return dict(
        name=name, fn=fn_name, n=n, d=d, dtype=dt, projection=projection,
        X=np.ascontiguousarray(u.astype(dt)), y=np.ascontiguousarray(y.astype(dt)), nu=2.5,
        theta=theta, theta0=theta0, lo=lo_b, hi=hi_b, amplitude=c0,
    )


def restart_points(name, lo, hi, count):
    """Uniform start points in log-bounds (gradmin.rs:22-24), shape [count, p]."""
    idx = CONFIGS[name][5]
    p = len(lo)
    u = splitmix64_uniform_fast(SEED_BASE + 1000 + idx, count * p).reshape(count, p)
    return np.log(lo)[None, :] + (np.log(hi) - np.log(lo))[None, :] * u


def candidates(name, m, d):
    idx = CONFIGS[name][5]
    return splitmix64_uniform_fast(SEED_BASE + 2000 + idx, m * d).reshape(m, d)
