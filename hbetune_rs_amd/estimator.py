"""Host-side mirror of the reference's GP adapter: `EstimatorGPR` / `SurrogateModelGPR` (src/core/gpr.rs:54-63, 215-450)
over the C ABI, with the O(n) scalar pieces the adapter keeps on the host: y-normalisation (src/core/ynormalize.rs),
the amplitude heuristic (gpr.rs:429-450), expected improvement (src/core/acquisition.rs:141-171) and summary statistics
(gpr.rs:114-177).  Same names, argument meaning and error behaviour as the traits in src/core/surrogate_model.rs:6-65,
so that tests read like tests/gpr_tests.rs.  All O(n^2 d)/O(n^3) work happens in libhbegp.so on the GPU.
"""
import math

import numpy as np

from . import gpr
from .synth import splitmix64_uniform_fast

FUDGE_MIN = 0.05  # ynormalize.rs:5


class BoundsError(ValueError):
    """bounded_value.rs:73-77 / gpr.rs:453-473"""

    def __init__(self, what, value, lo, hi):
        super().__init__(f"{what} {value} violated bounds [{lo}, {hi}] during model fitting")
        self.value, self.min, self.max = value, lo, hi


def _bounded(what, value, lo, hi):  # BoundedValue::new bounded_value.rs:14-20
    if not (lo <= value <= hi):
        raise BoundsError(what, value, lo, hi)
    return value


class YNormalize:
    """ynormalize.rs:7-288: linear or logarithmic projection of y into the normalised range and back."""

    def __init__(self, amplitude, expected, projection):
        self.amplitude, self.expected, self.projection = amplitude, expected, projection

    @staticmethod
    def new_project_into_normalized(y, projection="linear", known_optimum=None):  # :162-195
        y = np.asarray(y)
        dt = y.dtype.type
        minimum = dt(0) if projection == "linear" else dt(1)
        expected = y.min() - minimum  # guess_min :291-303
        if known_optimum is not None and dt(known_optimum) < expected:
            expected = dt(known_optimum)
        if projection == "linear":
            yn = y - expected
            amp = yn.mean()
            amp = amp if amp > 0 else dt(1)  # guess_amplitude :307-322
            return yn / amp + dt(FUDGE_MIN), YNormalize(amp, expected, projection)
        if projection == "logarithmic":
            yn = np.log(y - expected)
            amp = yn.mean()
            amp = amp if amp > 0 else dt(1)
            return yn / amp, YNormalize(amp, expected, projection)
        raise ValueError(projection)

    def project_into_normalized(self, y):  # :197-209
        y = np.asarray(y)
        if self.projection == "linear":
            return (y - self.expected) / self.amplitude + y.dtype.type(FUDGE_MIN)
        return np.log(y - self.expected) / self.amplitude

    def project_location_from_normalized(self, y):  # :214-224
        y = np.asarray(y)
        if self.projection == "linear":
            return (y - y.dtype.type(FUDGE_MIN)) * self.amplitude + self.expected
        return np.exp(y * self.amplitude) + self.expected

    def project_mean_from_normalized(self, mean, variance):  # :226-245
        mean, variance = np.asarray(mean), np.asarray(variance)
        if self.projection == "linear":
            return (mean - mean.dtype.type(FUDGE_MIN)) * self.amplitude + self.expected
        return np.exp(mean * self.amplitude + variance * self.amplitude ** 2 / 2) + self.expected

    def project_std_from_normalized(self, mean, variance):  # :247-265
        mean, variance = np.asarray(mean), np.asarray(variance)
        if self.projection == "linear":
            return np.sqrt(variance) * self.amplitude
        mu, s2 = mean * self.amplitude, variance * self.amplitude ** 2
        return np.sqrt(np.exp(mu * 2 + s2) * (np.exp(s2) - 1))  # logwarp::project_variance_from :112-119

    def project_cv_from_normalized(self, mean, variance):  # :267-286
        mean, variance = np.asarray(mean), np.asarray(variance)
        if self.projection == "linear":
            return np.sqrt(variance) * self.amplitude / ((mean - mean.dtype.type(FUDGE_MIN)) * self.amplitude + self.expected)
        return np.sqrt(np.exp(variance * self.amplitude ** 2) - 1)


def _norm_cdf(z):
    return 0.5 * math.erfc(-z / math.sqrt(2.0))


def _norm_pdf(z):
    return math.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)


def _norm_inverse_cdf(p, mu, sigma):
    from scipy.special import ndtri

    return mu + sigma * float(ndtri(p))


def expected_improvement(mean, std, fmin):
    """acquisition.rs:141-171"""
    assert math.isfinite(mean) and math.isfinite(std) and math.isfinite(fmin)
    if std <= 0.0 or abs(std) <= np.finfo(float).eps:  # ulps_eq!(std, 0.0): |std| <= f64::EPSILON (acquisition.rs:148)
        return -(mean - fmin) if mean < fmin else 0.0
    z = -(mean - fmin) / std
    ei = -(mean - fmin) * _norm_cdf(z) + std * _norm_pdf(z)
    assert math.isfinite(ei) and ei >= -1e-300
    return max(ei, 0.0)


class SummaryStatistics:
    """surrogate_model.rs:67-135"""

    def __init__(self, mean, std, cv, quartiles):
        self._mean, self._std, self._cv = mean, std, cv
        self.q1, self.q2, self.q3 = quartiles

    def mean(self):
        return self._mean

    def std(self):
        return self._std

    def cv(self):
        return self._cv

    def median(self):
        return self.q2

    def iqr(self):
        return self.q3 - self.q1


def estimate_amplitude(y, bounds=None):
    """gpr.rs:429-450 -> (start, lo, hi)"""
    y = np.asarray(y, dtype=np.float64)
    if bounds is None:
        hi = float((y ** 2).sum())
        ys = np.sort(y)
        q = float(ys[int(math.floor((len(ys) - 1) * 0.1))])  # Quantile1dExt::quantile_mut(0.1, Lower)
        lo = q * q * len(y)
        assert lo >= 0.0
        lo = lo if lo > 2e-5 else 2e-5
        lo, hi = lo / 2.0, hi * 2.0
    else:
        lo, hi = bounds
    start = math.exp((math.log(lo) + math.log(hi)) / 2.0)
    return _bounded("amplitude", start, lo, hi), lo, hi


class RNG:
    """Stand-in for the caller's RNG (random.rs:9-52): only `uniform` is needed by the fit (gradmin.rs:22-24)."""

    def __init__(self, seed):
        self.seed, self.count = int(seed), 0

    @staticmethod
    def new_with_seed(seed):
        return RNG(seed)

    def fork_random_state(self):
        self.count += 1
        return RNG(self.seed * 1000003 + self.count)

    def uniform(self, lo, hi, size):
        u = splitmix64_uniform_fast(self.seed + 7919 * self.count, size)
        self.count += 1
        return lo + (hi - lo) * u


class SurrogateModelGPR:
    """gpr.rs:54-63 + impl SurrogateModel (gpr.rs:71-213)."""

    def __init__(self, fitted, noise_bounds, amplitude_bounds, length_scale_bounds, y_norm, dtype):
        self.fitted = fitted  # gpr.FittedKernel: kernel parameters, alpha, k_inv on the device
        self.noise_bounds, self.amplitude_bounds, self.length_scale_bounds = noise_bounds, amplitude_bounds, length_scale_bounds
        self.y_norm = y_norm
        self.dtype = np.dtype(dtype)
        self.lml = fitted.lml

    def length_scales(self):  # gpr.rs:72-79
        return list(self.fitted.length_scale)

    def kernel(self):
        return dict(amplitude=self.fitted.amplitude, length_scale=self.fitted.length_scale, noise=self.fitted.noise)

    def predict_mean_a(self, x):  # gpr.rs:81-92
        y, _, _ = self.fitted.predict(np.asarray(x, dtype=self.dtype), want_variance=False)
        return self.y_norm.project_location_from_normalized(y)

    def predict_mean(self, x):  # surrogate_model.rs:42-45
        return self.predict_mean_a(np.asarray(x, dtype=self.dtype)[None, :])[0]

    def _predict_norm(self, x2d):
        m, v, n_warn = self.fitted.predict(np.asarray(x2d, dtype=self.dtype), want_variance=True)
        if n_warn:
            import sys

            print("Variances below 0 were predicted and will be corrected", file=sys.stderr)  # predict.rs:39-48
        return m, v

    def predict_confidence_bound(self, x, cb):  # gpr.rs:94-112
        m, v = self._predict_norm(np.asarray(x, dtype=self.dtype)[None, :])
        return self.y_norm.project_location_from_normalized(m + np.sqrt(v) * self.dtype.type(cb))[0]

    def predict_statistics(self, x):  # gpr.rs:114-177
        m, v = self._predict_norm(np.asarray(x, dtype=self.dtype)[None, :])
        std_n, mean_n = float(np.sqrt(v[0])), float(m[0])
        mean = self.y_norm.project_mean_from_normalized(m, v)[0]
        std = self.y_norm.project_std_from_normalized(m, v)[0]
        cv = self.y_norm.project_cv_from_normalized(m, v)[0]
        if abs(std_n) <= np.finfo(float).eps:  # abs_diff_eq!(std, 0.0): |std| <= f64::EPSILON (gpr.rs:133)
            qn = np.array([mean_n] * 3, dtype=self.dtype)
        else:
            qn = np.array([_norm_inverse_cdf(p, mean_n, std_n) for p in (0.25, 0.5, 0.75)], dtype=self.dtype)
        q = self.y_norm.project_location_from_normalized(qn)
        return SummaryStatistics(mean, std, cv, (q[0], q[1], q[2]))

    def predict_mean_ei_a(self, x, fmin):  # gpr.rs:179-212
        m, v = self._predict_norm(x)
        fmin_n = float(self.y_norm.project_into_normalized(np.array([fmin], dtype=self.dtype))[0])
        ei = np.array([expected_improvement(float(mi), float(math.sqrt(vi)), fmin_n) for mi, vi in zip(m, v)], dtype=self.dtype)
        return self.y_norm.project_location_from_normalized(m), ei

    # Batched forms of the scalar trait methods (the acquisition loops call these once per generation
    # instead of m single-point predicts, each of which reads all of K^-1).
    def predict_confidence_bound_a(self, x, cb):
        m, v = self._predict_norm(x)
        return self.y_norm.project_location_from_normalized(m + np.sqrt(v) * self.dtype.type(cb))

    def predict_mean_std_a(self, x):
        m, v = self._predict_norm(x)
        return self.y_norm.project_mean_from_normalized(m, v), self.y_norm.project_std_from_normalized(m, v)

    def predict_mean_ei(self, x, fmin):  # surrogate_model.rs:49-52
        mean, ei = self.predict_mean_ei_a(np.asarray(x, dtype=self.dtype)[None, :], fmin)
        return mean[0], ei[0]


def find_best_candidate_by_ei(candidates, model, fmin):
    """acquisition.rs:177-202 over a feature array: index, mean and EI of the candidate with maximal EI.
    One batched predict instead of a scalar `predict_mean_ei` per candidate.  Ties go to the LAST maximal element, as
    Rust's `Iterator::max_by` does; a NaN EI raises like the reference's panic."""
    means, eis = model.predict_mean_ei_a(np.asarray(candidates), fmin)
    return _pick(means, eis, _argmax_last(eis[None, :])[0])


def _argmax_last(eis2d):
    eis2d = np.asarray(eis2d)
    if np.isnan(eis2d).any():
        raise ValueError("EI should be comparable")  # acquisition.rs:193-195
    # index of the last maximum of every row
    rev = eis2d[:, ::-1]
    return eis2d.shape[1] - 1 - np.argmax(rev, axis=1)


def _pick(means, eis, i):
    return int(i), means[i], eis[i]


def acquire_by_mutation(candidates, model, fmin):
    """`MutationAcquisition::acquire` (acquisition.rs:86-116) after candidate generation: `candidates` is
    [n_parents, breadth, n_features] (the caller's `generate_nearby_samples`, projected into features); returns, per
    parent, the index of its best candidate and that candidate's mean and EI.  The whole generation is ONE batched
    predict instead of n_parents*breadth single-point predicts."""
    c = np.asarray(candidates)
    if c.ndim != 3:
        raise ValueError("candidates must be [n_parents, breadth, n_features]")
    npar, breadth, nf = c.shape
    means, eis = model.predict_mean_ei_a(c.reshape(npar * breadth, nf), fmin)
    means, eis = means.reshape(npar, breadth), eis.reshape(npar, breadth)
    idx = _argmax_last(eis)
    rows = np.arange(npar)
    return idx, means[rows, idx], eis[rows, idx]


class FitnessOperator:
    """`FitnessOperator(model, space, fitness_via)` (minimize.rs:653-678) over feature arrays.  The reference predicts the
    mean of ONE individual per `get_fitness` call -- twice per comparison inside `sort_by` / `select_next_population`
    (minimize.rs:509-514, 566-575), each an O(n^2) single-point predict.  Here the fitness of a whole population is ONE
    batched `predict_mean_a`; comparisons then run on the cached values with the same `partial_cmp` semantics."""

    def __init__(self, model, fitness_via="prediction"):
        if fitness_via not in ("prediction", "observation"):
            raise ValueError(fitness_via)
        self.model, self.fitness_via = model, fitness_via

    def get_fitness(self, features, observations=None):
        """fitness of every individual: predicted mean at its features, or its observation (minimize.rs:657-668)."""
        if self.fitness_via == "prediction":
            return np.asarray(self.model.predict_mean_a(np.asarray(features)))
        if observations is None:
            raise ValueError("individual has no observation")
        return np.asarray(observations)

    @staticmethod
    def compare(fa, fb):
        """`a.partial_cmp(&b)` (minimize.rs:670-677): -1 / 0 / 1, or None when not comparable (NaN)."""
        if math.isnan(fa) or math.isnan(fb):
            return None
        return int(fa > fb) - int(fa < fb)

    def sort_population(self, features, observations=None):
        """indices of the population sorted by fitness, best (lowest) first: `population.sort_by(compare)` in
        `resize_population` (minimize.rs:509-514; Rust's sort_by is stable).  Raises where the reference panics."""
        f = self.get_fitness(features, observations)
        if np.isnan(f).any():
            raise ValueError("individuals are comparable")
        return np.argsort(f, kind="stable"), f

    def select_next_population(self, parent_features, offspring_features, parent_obs=None, offspring_obs=None):
        """`select_next_population` (minimize.rs:722-747): each offspring competes against its one parent and is kept unless
        `compare(parent, offspring) == Some(Less)` (so an incomparable pair keeps the offspring).  Returns a boolean array:
        True where the offspring is selected.  Two batched predicts for the whole generation."""
        fp = self.get_fitness(parent_features, parent_obs)
        fo = self.get_fitness(offspring_features, offspring_obs)
        with np.errstate(invalid="ignore"):
            return ~(fp < fo)


def find_best_individual_by_confidence_bound(features, model, confidence_bound):
    """minimize.rs:680-714 over a feature array: the individual with the lowest confidence bound (the FIRST of several
    minimal ones: the reference replaces its suggestion only on a strictly lower bound) and the predicted mean there.  One
    batched `predict_confidence_bound_a` + one `predict_mean` instead of one single-point predict per individual."""
    features = np.asarray(features)
    if features.ndim != 2 or features.shape[0] < 1:
        raise ValueError("should have at least one individual")
    ucb = np.asarray(model.predict_confidence_bound_a(features, confidence_bound))
    best = 0
    for i in range(1, len(ucb)):
        if ucb[i] < ucb[best]:
            best = i
    return best, model.predict_mean(features[best])


class EstimatorGPR:
    """gpr.rs:215-400: defaults, builders, estimate(), extend()."""

    def __init__(self, n_features, ctx=None):  # Estimator::new(space) gpr.rs:219-236
        self._noise_bounds = (1e-5, 1e5)
        self._length_scale_bounds = [(1e-3, 1e3)] * n_features
        self._n_restarts_optimizer = 2
        self._matern_nu = 2.5
        self._amplitude_bounds = None
        self._y_projection = "linear"
        self._known_optimum = None
        self.ctx = ctx

    @staticmethod
    def new(space_or_dims, ctx=None):
        n = space_or_dims if isinstance(space_or_dims, int) else len(space_or_dims)
        return EstimatorGPR(n, ctx)

    # builders gpr.rs:351-400
    def noise_bounds(self, lo, hi):
        self._noise_bounds = (lo, hi)
        return self

    def length_scale_bounds(self, bounds):
        self._length_scale_bounds = list(bounds)
        return self

    def n_restarts_optimizer(self, n):
        self._n_restarts_optimizer = n
        return self

    def matern_nu(self, nu):
        self._matern_nu = nu
        return self

    def amplitude_bounds(self, bounds):
        self._amplitude_bounds = bounds
        return self

    def y_projection(self, projection):
        self._y_projection = projection
        return self

    def known_optimum(self, value):
        self._known_optimum = value
        return self

    def _theta_and_bounds(self, prior, y_train):  # get_kernel_or_default gpr.rs:402-427
        if prior is not None:
            lo = np.array([prior.noise_bounds[0], prior.amplitude_bounds[0]] + [b[0] for b in prior.length_scale_bounds])
            hi = np.array([prior.noise_bounds[1], prior.amplitude_bounds[1]] + [b[1] for b in prior.length_scale_bounds])
            return prior.fitted.theta.copy(), lo, hi
        amp, a_lo, a_hi = estimate_amplitude(y_train, self._amplitude_bounds)  # gpr.rs:262
        noise = _bounded("noise level", 1.0, *self._noise_bounds)
        ells = [_bounded("length scale", math.exp((math.log(lo) + math.log(hi)) / 2.0), lo, hi) for lo, hi in self._length_scale_bounds]
        theta0 = np.log(np.array([noise, amp] + ells))
        lo = np.array([self._noise_bounds[0], a_lo] + [b[0] for b in self._length_scale_bounds])
        hi = np.array([self._noise_bounds[1], a_hi] + [b[1] for b in self._length_scale_bounds])
        return theta0, lo, hi

    def estimate(self, x, y, prior, rng):  # gpr.rs:238-291
        x = np.asarray(x)
        y = np.asarray(y, dtype=x.dtype)
        assert y.shape == (x.shape[0],), f"expected y values for {x.shape[0]} observations"
        y_train, y_norm = YNormalize.new_project_into_normalized(y, self._y_projection, self._known_optimum)
        theta0, lo, hi = self._theta_and_bounds(prior, y_train)
        fork = rng.fork_random_state()  # gpr.rs:276
        p = len(theta0)
        starts = None
        if self._n_restarts_optimizer > 0:  # gradmin.rs:22-24: uniform in log-bounds
            u = fork.uniform(0.0, 1.0, self._n_restarts_optimizer * p).reshape(self._n_restarts_optimizer, p)
            starts = np.log(lo)[None, :] + (np.log(hi) - np.log(lo))[None, :] * u
        # a prior hands over its whole kernel -- theta, bounds AND the Matern nu (prior.kernel.clone(), gpr.rs:407-409);
        # the estimator's own nu only configures the default kernel
        nu = prior.fitted.nu if prior is not None else self._matern_nu
        fitted = gpr.FittedKernel.new(x, y_train.astype(x.dtype), theta0, lo, hi, starts, nu=nu, ctx=self.ctx)
        return SurrogateModelGPR(fitted, (lo[0], hi[0]), (lo[1], hi[1]), list(zip(lo[2:], hi[2:])), y_norm, x.dtype)

    def extend(self, x, y, prior, rng=None):  # gpr.rs:293-337
        x = np.asarray(x)
        y = np.asarray(y, dtype=x.dtype)
        assert y.shape == (x.shape[0],), f"expected y values for {x.shape[0]} observations"
        y_train, y_norm = YNormalize.new_project_into_normalized(y, self._y_projection, self._known_optimum)
        lo = np.array([prior.noise_bounds[0], prior.amplitude_bounds[0]] + [b[0] for b in prior.length_scale_bounds])
        hi = np.array([prior.noise_bounds[1], prior.amplitude_bounds[1]] + [b[1] for b in prior.length_scale_bounds])
        if prior.fitted.dtype == x.dtype:  # the kernel (incl. its nu) is the prior's (prior.kernel.clone(), gpr.rs:318-323)
            # the caller appends its validation samples to the rows the prior was built on (minimize.rs:629-644): the engine
            # reuses the prior's factorisation for the unchanged leading rows (falls back by itself when they differ)
            fitted = prior.fitted.extend_with(x, y_train.astype(x.dtype), ctx=self.ctx)
        else:
            fitted = gpr.FittedKernel.extend(x, y_train.astype(x.dtype), prior.fitted.theta, lo, hi, nu=prior.fitted.nu, ctx=self.ctx)
        return SurrogateModelGPR(fitted, prior.noise_bounds, prior.amplitude_bounds, prior.length_scale_bounds, y_norm, x.dtype)
