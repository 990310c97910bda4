"""CPU: host-side adapter pieces restated from the reference (ynormalize.rs, acquisition.rs::expected_improvement,
gpr.rs::estimate_amplitude) against values computed independently."""
import math

import numpy as np
import pytest

from hbetune_rs_amd import estimator as E


def test_ynormalize_linear_roundtrip_and_moments():
    y = np.array([1.0, 1.8, 2.2, 3.0])
    yn, cfg = E.YNormalize.new_project_into_normalized(y, "linear")
    assert cfg.expected == 1.0 and cfg.amplitude == pytest.approx(1.0)  # (y - min).mean() = 1
    np.testing.assert_allclose(yn, (y - 1.0) / 1.0 + 0.05)
    np.testing.assert_allclose(cfg.project_location_from_normalized(yn), y)
    np.testing.assert_allclose(cfg.project_into_normalized(y), yn)
    np.testing.assert_allclose(cfg.project_std_from_normalized(yn, np.full(4, 0.04)), 0.2 * cfg.amplitude)


def test_ynormalize_logarithmic_matches_lognormal_formulas():
    y = np.array([3.0, 10.0, 30.0, 100.0])
    yn, cfg = E.YNormalize.new_project_into_normalized(y, "logarithmic")
    assert cfg.expected == 2.0  # min - 1
    np.testing.assert_allclose(cfg.project_location_from_normalized(yn), y)
    mu, s2 = np.array([0.3]), np.array([0.2])
    a = cfg.amplitude
    np.testing.assert_allclose(cfg.project_mean_from_normalized(mu, s2), np.exp(mu * a + s2 * a * a / 2) + cfg.expected)
    np.testing.assert_allclose(cfg.project_cv_from_normalized(mu, s2), np.sqrt(np.exp(s2 * a * a) - 1))


def test_known_optimum_lowers_expected():
    y = np.array([1.0, 2.0])
    _, cfg = E.YNormalize.new_project_into_normalized(y, "linear", known_optimum=0.0)
    assert cfg.expected == 0.0


def test_expected_improvement_cases():
    from scipy.stats import norm

    # acquisition.rs:141-171: zero std -> pure difference / 0
    assert E.expected_improvement(1.0, 0.0, 2.0) == 1.0
    assert E.expected_improvement(3.0, 0.0, 2.0) == 0.0
    for mean, std, fmin in [(0.5, 0.3, 0.4), (1.0, 2.0, -1.0), (-0.2, 0.1, 0.0)]:
        z = -(mean - fmin) / std
        want = -(mean - fmin) * norm.cdf(z) + std * norm.pdf(z)
        assert E.expected_improvement(mean, std, fmin) == pytest.approx(want, rel=1e-12)


def test_expected_improvement_zero_std_boundary_is_f64_epsilon():
    # ulps_eq!(std, 0.0) (acquisition.rs:148) is |std| <= f64::EPSILON: at and below it the degenerate branch runs,
    # just above it the z-score branch does
    eps = np.finfo(float).eps
    assert E.expected_improvement(1.0, eps, 2.0) == 1.0
    assert E.expected_improvement(1.0, eps / 2, 2.0) == 1.0
    assert E.expected_improvement(3.0, eps, 2.0) == 0.0
    above = E.expected_improvement(1.0, 4 * eps, 2.0)
    assert above == pytest.approx(1.0, rel=1e-12)  # same value in the limit, but through cdf/pdf
    assert E.expected_improvement(2.0, 1e-9, 2.0) > 0.0  # mean == fmin: only the z-score branch gives std * pdf(0)


def test_estimate_amplitude():
    y = np.array([0.05, 0.5, 1.0, 2.45])
    start, lo, hi = E.estimate_amplitude(y)
    assert hi == pytest.approx(2 * (y ** 2).sum())
    assert lo == pytest.approx(max(0.05 ** 2 * 4, 2e-5) / 2)
    assert start == pytest.approx(math.sqrt(lo * hi))
    assert E.estimate_amplitude(y, (0.1, 4.0))[1:] == (0.1, 4.0)


def test_bounds_error_like_reference():
    est = E.EstimatorGPR.new(1).noise_bounds(2.0, 3.0)  # default start 1.0 is outside -> Error::NoiseBounds (gpr.rs:411-412)
    with pytest.raises(E.BoundsError):
        est._theta_and_bounds(None, np.array([0.1, 1.0]))


class _FakeModel:
    """Stands in for SurrogateModelGPR: EI is a fixed function of the first feature (no GPU needed)."""

    def __init__(self):
        self.calls = 0

    def predict_mean_ei_a(self, x, fmin):
        self.calls += 1
        x = np.asarray(x)
        ei = np.round(np.sin(7.0 * x[:, 0]) ** 2, 1)  # coarse values -> plenty of ties
        return x.sum(axis=1), ei


def test_batched_acquisition_matches_scalar_loop_and_rust_tie_rule():
    rng = np.random.default_rng(3)
    cand = rng.random((6, 25, 3))
    model = _FakeModel()
    idx, mean, ei = E.acquire_by_mutation(cand, model, fmin=0.0)
    assert model.calls == 1  # the whole generation in one predict
    for p in range(6):
        # acquisition.rs:177-202 with Iterator::max_by: the LAST of several maximal elements wins
        m_p, e_p = _FakeModel().predict_mean_ei_a(cand[p], 0.0)
        best = max(range(25), key=lambda i: (e_p[i], i))
        assert idx[p] == best and ei[p] == e_p[best] and mean[p] == m_p[best]
        i1, m1, e1 = E.find_best_candidate_by_ei(cand[p], _FakeModel(), 0.0)
        assert (i1, m1, e1) == (best, m_p[best], e_p[best])


def test_batched_acquisition_rejects_nan_ei():
    class Bad(_FakeModel):
        def predict_mean_ei_a(self, x, fmin):
            m, e = super().predict_mean_ei_a(x, fmin)
            e[0] = np.nan
            return m, e

    with pytest.raises(ValueError):
        E.find_best_candidate_by_ei(np.zeros((4, 3)), Bad(), 0.0)


# ---- vectors the reference's own ynormalize.rs tests hold (tests/golden/reference_kats.json, data only) --------------
import json
import os

_YK = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))["ynormalize"]


def _log_norm(expected=0.0, amplitude=1.0):
    return E.YNormalize(amplitude, expected, "logarithmic")


def test_ref_logwarp_project_mean_from():  # ynormalize.rs:48-75
    k = _YK["logwarp_project_mean_from"]
    m, s = np.array(k["logmean"]), np.array(k["logstd"])
    got = _log_norm().project_mean_from_normalized(m, s * s)
    np.testing.assert_allclose(got, np.exp(k["expected_exponents"]), rtol=0, atol=k["epsilon"])


def test_ref_logwarp_project_variance():  # ynormalize.rs:125-147
    k = _YK["logwarp_project_variance"]
    m, s = np.array(k["logmean"]), np.array(k["logstd"])
    n = _log_norm()
    np.testing.assert_allclose(n.project_std_from_normalized(m, s * s), n.project_mean_from_normalized(m, s * s) * np.sqrt(np.exp(s * s) - 1),
                               rtol=0, atol=k["epsilon"])
    np.testing.assert_allclose(n.project_cv_from_normalized(m, s * s), np.sqrt(np.exp(s * s) - 1), rtol=0, atol=1e-12)  # :150-156


def test_ref_lognormal_moments_from_data():  # ynormalize.rs:78-112 (own sample: the reference's RNG stream is not available)
    k = _YK["lognormal_from_data"]
    data = np.exp(np.random.default_rng(83229).normal(k["mu"], k["sigma"], k["count"]))
    logmean, logvar = np.log(data).mean(), np.log(data).var()
    n = _log_norm()
    mean = n.project_mean_from_normalized(np.array([logmean]), np.array([logvar]))[0]
    std = n.project_std_from_normalized(np.array([logmean]), np.array([logvar]))[0]
    assert abs(mean - data.mean()) <= k["mean_max_relative"] * max(abs(mean), data.mean())
    assert abs(std - data.std()) <= k["std_max_relative"] * max(abs(std), data.std())


@pytest.mark.parametrize("case", _YK["inverse"]["cases"])
def test_ref_projection_has_an_inverse(case):  # ynormalize.rs:324-383
    y = np.array(case["input"])
    yn, norm = E.YNormalize.new_project_into_normalized(y, case["projection"], case["known_optimum"])
    eps = _YK["inverse"]["epsilon"]
    np.testing.assert_allclose(norm.project_location_from_normalized(yn), y, rtol=0, atol=eps)
    np.testing.assert_allclose(norm.project_into_normalized(y), yn, rtol=0, atol=eps)
    if case["known_optimum"] is not None:
        assert norm.expected == min(y.min() - (0.0 if case["projection"] == "linear" else 1.0), case["known_optimum"])  # guess_min :291-303


def test_ref_linear_can_handle_variance():  # ynormalize.rs:385-399
    k = _YK["linear_variance"]
    norm = E.YNormalize(k["amplitude"], k["expected"], "linear")
    np.testing.assert_allclose(norm.project_std_from_normalized(np.zeros(3), np.array(k["variance"])), k["std"], rtol=0, atol=k["epsilon"])


@pytest.mark.parametrize("case", _YK["logarithmic_mean"]["cases"])
def test_ref_logarithmic_project_mean(case):  # ynormalize.rs:401-463
    norm = E.YNormalize(case["amplitude"], case["expected"], "logarithmic")
    got = norm.project_mean_from_normalized(np.array([case["mean"]]), np.array([case["std"] ** 2]))[0]
    assert abs(got - case["want"]) <= _YK["logarithmic_mean"]["epsilon"] * max(1.0, abs(case["want"]) * 1e-2)  # abs_diff_eq 1e-7 at e..e^6.5


@pytest.mark.parametrize("projection", ["linear", "logarithmic"])
def test_ref_statistics_survive_the_projection(projection):  # ynormalize.rs:465-521
    # The log-normal moment match is a property of the sample: the reference's tolerances (1 % / 10 %) hold for about a third
    # of all 500-point samples, its seed (903282318 on its own RNG) picks one of them -- ours (903282322 on numpy's) too.
    k = _YK["statistics"]
    data = np.exp(np.random.default_rng(903282322).normal(k["mu"], k["sigma"], k["count"]))
    want_mean, want_std = data.mean(), data.std()
    t, norm = E.YNormalize.new_project_into_normalized(data, projection)
    tm, tv = np.array([t.mean()]), np.array([t.var()])
    mean = norm.project_mean_from_normalized(tm, tv)[0]
    std = norm.project_std_from_normalized(tm, tv)[0]
    cv = norm.project_cv_from_normalized(tm, tv)[0]
    r = k[projection]
    rel = lambda a, b, tol: abs(a - b) <= tol * max(abs(a), abs(b))  # relative_eq!(max_relative)
    assert rel(mean, want_mean, r["mean_ratio"]) and rel(std, want_std, r["std_ratio"])
    assert rel(cv, want_std / want_mean, math.hypot(r["mean_ratio"], r["std_ratio"]))


def test_predict_statistics_zero_std_boundary_is_f64_epsilon():
    # gpr.rs:133: abs_diff_eq!(std, 0.0) -> |std| <= f64::EPSILON takes the degenerate branch (all quartiles = mean)
    class FakeFitted:
        lml = 0.0

        def __init__(self, var):
            self.var = var

        def predict(self, x, want_variance=True):
            return np.array([0.7]), np.array([self.var]), 0

    eps = np.finfo(float).eps
    for var, degenerate in [(0.0, True), (eps * eps, True), ((4 * eps) ** 2, False), (0.04, False)]:
        m = E.SurrogateModelGPR(FakeFitted(var), (1e-5, 1e5), (0.1, 10.0), [(1e-3, 1e3)], E.YNormalize(2.0, 1.0, "linear"), np.float64)
        st = m.predict_statistics(np.array([0.5]))
        assert st.mean() == pytest.approx((0.7 - 0.05) * 2.0 + 1.0)
        if degenerate:
            assert st.q1 == st.q2 == st.q3 == st.mean()
        else:
            assert st.q1 < st.q2 < st.q3 and st.iqr() == pytest.approx(2 * 0.6744897501960817 * math.sqrt(var) * 2.0, rel=1e-9)


# ---- batched forms of the caller's scalar-predict loops, host logic on a fake model ---------------
class _CbModel:
    """mean = first feature, std = second feature (no GPU needed)."""

    def __init__(self):
        self.calls = []

    def predict_mean_a(self, x):
        self.calls.append(("mean_a", len(x)))
        return np.asarray(x)[:, 0].copy()

    def predict_mean(self, x):
        self.calls.append(("mean", 1))
        return float(np.asarray(x)[0])

    def predict_confidence_bound(self, x, cb):
        self.calls.append(("cb", 1))
        return float(x[0] + cb * x[1])

    def predict_confidence_bound_a(self, x, cb):
        self.calls.append(("cb_a", len(x)))
        x = np.asarray(x)
        return x[:, 0] + cb * x[:, 1]


def test_find_best_individual_by_confidence_bound_matches_the_reference_loop():
    rng = np.random.default_rng(4)
    feats = np.round(rng.random((40, 2)), 1)  # coarse values: ties
    for cb in (-1.0, 0.0, 2.0):
        m = _CbModel()
        got_i, got_y = E.find_best_individual_by_confidence_bound(feats, m, cb)
        assert [c[0] for c in m.calls] == ["cb_a", "mean"]  # one batched predict + the final mean
        # minimize.rs:680-714, literally
        ref = _CbModel()
        sug, sug_ucb = 0, ref.predict_confidence_bound(feats[0], cb)
        for i in range(1, len(feats)):
            c = ref.predict_confidence_bound(feats[i], cb)
            if c < sug_ucb:
                sug, sug_ucb = i, c
        assert got_i == sug and got_y == ref.predict_mean(feats[sug])
    with pytest.raises(ValueError):
        E.find_best_individual_by_confidence_bound(np.zeros((0, 2)), _CbModel(), 1.0)


def test_fitness_operator_batched_sort_and_selection():
    rng = np.random.default_rng(9)
    feats = np.round(rng.random((30, 2)), 1)
    obs = np.round(rng.random(30), 1)
    m = _CbModel()
    op = E.FitnessOperator(m, "prediction")
    order, f = op.sort_population(feats)
    assert m.calls == [("mean_a", 30)]
    # the reference: population.sort_by(|a, b| fitness.compare(a, b)) with a stable sort (minimize.rs:509-514)
    import functools

    ref = sorted(range(30), key=functools.cmp_to_key(lambda a, b: E.FitnessOperator.compare(feats[a, 0], feats[b, 0])))
    assert order.tolist() == ref
    order_o, _ = E.FitnessOperator(m, "observation").sort_population(feats, obs)
    assert order_o.tolist() == sorted(range(30), key=lambda i: (obs[i], i))
    # select_next_population (minimize.rs:722-747): parent kept only when strictly better; NaN keeps the offspring
    parents, offspring = feats[:15], feats[15:].copy()
    offspring[3, 0] = np.nan
    keep_off = op.select_next_population(parents, offspring)
    for i in range(15):
        cmp = E.FitnessOperator.compare(parents[i, 0], offspring[i, 0])
        assert keep_off[i] == (cmp != -1)
    with pytest.raises(ValueError):
        op.sort_population(offspring)  # NaN fitness: "individuals are comparable" panics in the reference
    assert E.FitnessOperator.compare(1.0, float("nan")) is None
