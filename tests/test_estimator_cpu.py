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
