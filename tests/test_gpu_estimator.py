"""GPU: behavioural suites of the reference's tests/gpr_tests.rs, through the EstimatorGPR / SurrogateModelGPR mirror."""
import numpy as np
import pytest

from hbetune_rs_amd import estimator as E
from hbetune_rs_amd.estimator import EstimatorGPR, RNG

pytestmark = pytest.mark.gpu


class SimpleModel:  # gpr_tests.rs:11-35 (1-D space on [lo, hi])
    def __init__(self, model, lo=0.0, hi=1.0):
        self.model, self.lo, self.hi = model, lo, hi

    def feat(self, x):
        return np.array([(x - self.lo) / (self.hi - self.lo)])

    def predict(self, x):
        return self.model.predict_mean(self.feat(x))

    def uncertainty(self, x):
        return self.model.predict_statistics(self.feat(x)).std()


@pytest.fixture(scope="module")
def density_model():  # gpr_tests.rs:83-91
    xs = np.array([0.1, 0.5, 0.5, 0.9])[:, None]
    ys = np.array([1.0, 1.8, 2.2, 3.0])
    return SimpleModel(EstimatorGPR.new(1).estimate(xs, ys, None, RNG.new_with_seed(123)))


def test_should_roughly_fit_the_data(density_model):  # :93-103
    got = [density_model.predict(x) for x in (0.1, 0.5, 0.9)]
    np.testing.assert_allclose(got, [1.0, 2.0, 3.0], atol=0.1)


def test_reasonable_interpolation_and_conservative_extrapolation(density_model):  # :105-117
    assert abs(density_model.predict(0.3) - 1.5) <= 0.1
    assert abs(density_model.predict(0.7) - 2.5) <= 0.1
    assert abs(density_model.predict(0.0) - 0.9) <= 0.1
    assert abs(density_model.predict(1.0) - 3.1) <= 0.1


def test_uncertainty_ordering(density_model):  # :119-129
    assert abs(density_model.uncertainty(0.1) - density_model.uncertainty(0.9)) <= 0.05
    assert density_model.uncertainty(0.5) < density_model.uncertainty(0.1)


@pytest.fixture(scope="module")
def unsampled_model():  # gpr_tests.rs:135-146
    xs = np.array([0.3, 0.5, 0.7])[:, None]
    ys = np.array([1.0, 2.0, 1.5])
    est = EstimatorGPR.new(1).noise_bounds(1e-5, 1e0).length_scale_bounds([(0.1, 1.0)])
    return SimpleModel(est.estimate(xs, ys, None, RNG.new_with_seed(9372)))


def test_unsampled_regions(unsampled_model):  # :148-168
    m = unsampled_model
    for x in (0.3, 0.5, 0.7):
        assert m.uncertainty(x) < 0.01
    for x in (0.4, 0.6, 0.0, 1.0):
        assert m.uncertainty(x) > 10.0 * m.uncertainty(0.3)


def test_works_in_1d():  # gpr_tests.rs:172-224
    xs_nat = np.linspace(-2.0, 2.0, 5)
    ys = xs_nat ** 2
    feats = ((xs_nat + 2.0) / 4.0)[:, None]
    est = EstimatorGPR.new(1).length_scale_bounds([(1e-2, 1e1)]).noise_bounds(1e-2, 1e1)
    model = est.estimate(feats, ys, None, RNG.new_with_seed(4531))

    def check(points):
        for x in points:
            st = model.predict_statistics(np.array([(x + 2.0) / 4.0]))
            expected = x * x
            assert expected - 0.6 * st.std() < st.mean() < expected + st.std(), (x, st.mean(), st.std())

    check(xs_nat)
    check([-1.5, -0.5, 1.5])


def test_2d_sphere_with_noise_and_extend_and_ei():
    # describe_2d in spirit (gpr_tests.rs:293-361): 49-point grid on the 2-D sphere, noise 0.1
    g = np.linspace(0.0, 1.0, 7)
    X = np.array([[a, b] for a in g for b in g])
    nat = X * 4.0 - 2.0
    rng = np.random.default_rng(5)
    y = (nat ** 2).sum(axis=1) + 0.1 * rng.standard_normal(len(X))
    est = EstimatorGPR.new(2)
    model = est.estimate(X, y, None, RNG.new_with_seed(77))
    pred = model.predict_mean_a(X)
    rmse = float(np.sqrt(np.mean((pred - (nat ** 2).sum(axis=1)) ** 2)))
    assert rmse <= 0.15
    assert len(model.length_scales()) == 2
    # extend: more data, same hyper-parameters (gpr.rs:293-337)
    X2 = np.vstack([X, [[0.33, 0.41]]])
    y2 = np.append(y, ((X2[-1] * 4 - 2) ** 2).sum())
    ext = est.extend(X2, y2, model)
    np.testing.assert_allclose(ext.length_scales(), model.length_scales())
    # EI is non-negative and larger near the optimum than at the corner
    mean, ei = model.predict_mean_ei_a(np.array([[0.5, 0.5], [0.0, 0.0]]), float(y.min()))
    assert np.all(ei >= 0) and ei[0] > ei[1]
    # warm start from a prior model reuses its parameters and bounds (gpr.rs:407-409)
    again = est.estimate(X, y, model, RNG.new_with_seed(78))
    assert again.lml >= model.lml - 1e-6
    # confidence bound moves with cb
    lo = model.predict_confidence_bound(np.array([0.1, 0.9]), -1.0)
    hi = model.predict_confidence_bound(np.array([0.1, 0.9]), 1.0)
    assert lo < hi


def test_batched_forms_match_scalar_calls(density_model):
    m = density_model.model
    xs = np.linspace(0.05, 0.95, 7)[:, None]
    cb_a = m.predict_confidence_bound_a(xs, 1.5)
    cb_s = [m.predict_confidence_bound(x, 1.5) for x in xs]
    np.testing.assert_allclose(cb_a, cb_s, rtol=1e-12, atol=1e-12)
    mean_a, std_a = m.predict_mean_std_a(xs)
    st = [m.predict_statistics(x) for x in xs]
    np.testing.assert_allclose(mean_a, [s.mean() for s in st], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(std_a, [s.std() for s in st], rtol=1e-12, atol=1e-12)


def test_batched_acquisition_matches_the_reference_scalar_loop(density_model):
    """acquisition.rs:86-116, 177-202: per parent, the candidate with maximal EI -- one batched predict for the whole
    generation against the reference's loop of single-point `predict_mean_ei` calls."""
    m = density_model.model
    rng = np.random.default_rng(11)
    cand = rng.random((5, 20, 1))
    fmin = 0.3
    idx, mean, ei = E.acquire_by_mutation(cand, m, fmin)
    for p in range(5):
        scalar = [m.predict_mean_ei(c, fmin) for c in cand[p]]
        e = np.array([s[1] for s in scalar])
        best = max(range(len(e)), key=lambda i: (e[i], i))  # Iterator::max_by keeps the last maximum
        assert idx[p] == best
        np.testing.assert_allclose(mean[p], scalar[best][0], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(ei[p], scalar[best][1], rtol=1e-8, atol=1e-12)
