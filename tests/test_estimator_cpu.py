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
