"""mogp_emulator_amd.validation (consumer of predict(full_cov=True)) against the reference's
golden vectors: predictions and the pivoted factorisation of the predictive covariance both come from the device."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import mogp_emulator_amd as M
from mogp_emulator_amd import validation as V
from mogp_emulator_amd.Priors import GPPriors
from conftest import load_golden

pytestmark = pytest.mark.gpu
KERNELS = ["SquaredExponential", "Matern52"]


def build(g, kern, mode, multi):
    nug = {"fixed": 1.e-4, "fit": "fit"}[mode]
    pri = GPPriors(n_corr=2, nugget_type=mode)
    theta = g["%s_%s_theta" % (kern, mode)]
    if multi:
        gp = M.MultiOutputGP_GPU(g["X"], g["T"], kernel=kern, nugget=nug, priors=pri)
        gp.fit(np.tile(theta, (3, 1)))
    else:
        gp = M.GaussianProcessGPU(g["X"], g["T"][0], kernel=kern, nugget=nug, priors=pri)
        gp.fit(theta)
    return gp


@pytest.mark.parametrize("kern", KERNELS)
@pytest.mark.parametrize("mode", ["fixed", "fit"])
def test_single_emulator_vs_reference(kern, mode):
    g = load_golden("validation.npz")
    pre = "%s_%s_" % (kern, mode)
    gp = build(g, kern, mode, False)
    e, P = V.standard_errors(gp, g["Xv"], g["Tv"][0])
    assert list(P) == list(g[pre + "std_P"])
    assert_allclose(e, g[pre + "std_err"], rtol=1e-6, atol=1e-8)
    e, P = V.pivoted_errors(gp, g["Xv"], g["Tv"][0])
    assert list(P) == list(g[pre + "piv_P"])
    assert_allclose(e, g[pre + "piv_err"], rtol=1e-6, atol=1e-7)
    assert_allclose(V.mahalanobis(gp, g["Xv"], g["Tv"][0]), g[pre + "mahal"], rtol=1e-7)
    assert_allclose(V.mahalanobis(gp, g["Xv"], g["Tv"][0], scaled=True), g[pre + "mahal_scaled"], rtol=1e-7)
    d = V.generate_mahal_dist(gp, g["Xv"])
    assert [d.kwds["dfn"], d.kwds["dfd"], d.kwds["scale"]] == list(g[pre + "dist_args"])
    assert_allclose(V.compute_errors(gp, g["Xv"], g["Tv"][0], "pivot")[0], e)


@pytest.mark.parametrize("kern", KERNELS)
@pytest.mark.parametrize("mode", ["fixed", "fit"])
def test_multi_output_vs_reference(kern, mode):
    g = load_golden("validation.npz")
    pre = "%s_%s_" % (kern, mode)
    mo = build(g, kern, mode, True)
    se = V.standard_errors(mo, g["Xv"], g["Tv"])
    pe = V.pivoted_errors(mo, g["Xv"], g["Tv"])
    assert len(se) == len(pe) == 3
    for k in range(3):
        assert list(se[k][1]) == list(g[pre + "mo_std_P"][k]) and list(pe[k][1]) == list(g[pre + "mo_piv_P"][k])
        assert_allclose(se[k][0], g[pre + "mo_std_err"][k], rtol=1e-6, atol=1e-8)
        assert_allclose(pe[k][0], g[pre + "mo_piv_err"][k], rtol=1e-6, atol=1e-7)
    assert_allclose(V.mahalanobis(mo, g["Xv"], g["Tv"]), g[pre + "mo_mahal"], rtol=1e-7)
    assert_allclose(V.mahalanobis(mo, g["Xv"], g["Tv"], scaled=True), g[pre + "mo_mahal_scaled"], rtol=1e-7)


def test_argument_checks_follow_the_reference():
    g = load_golden("validation.npz")
    gp = build(g, "Matern52", "fixed", False)
    mo = build(g, "Matern52", "fixed", True)
    with pytest.raises(AssertionError):
        V.standard_errors(gp, g["Xv"], g["Tv"])                 # 2-D targets for a single emulator, validation.py:473-477
    with pytest.raises(AssertionError):
        V.pivoted_errors(mo, g["Xv"], g["Tv"][0])               # 1-D targets for a multi-output emulator, :478-482
    with pytest.raises(AssertionError):
        V.standard_errors(gp, g["Xv"][:-1], g["Tv"][0])
    with pytest.raises(AssertionError):
        V.mahalanobis(3.0, g["Xv"], g["Tv"][0])
    with pytest.raises(TypeError):
        V.generate_mahal_dist(3.0, g["Xv"])
    with pytest.raises(ValueError):
        V.compute_errors(gp, g["Xv"], g["Tv"][0], "nonsense")


# ---- the reference's own cases (tests/test_validation.py:8-175): predictions replaced by fixed numbers, so that only the
# ---- error definitions and the (device) pivoted factorisation of the 3 x 3 covariance are exercised -------------------
targets1 = np.array([0.5, 2.1, 2.8])
targets2 = np.array([2.5, 2.9, 3.5])
mean1 = np.array([1.0, 2.0, 3.0])
mean2 = np.array([2.0, 3.0, 4.0])
err1, err2 = mean1 - targets1, mean2 - targets2
unc1 = np.array([[0.1, 0.05, 0.02], [0.05, 0.2, 0.01], [0.02, 0.01, 0.15]])
idx1 = np.array([1, 2, 0])


def mock_predict(self, testing, unc=True, deriv=False, include_nugget=True, full_cov=False):
    return M.PredictResult(mean=mean1, unc=unc1 if full_cov else np.diag(unc1), deriv=None)


def mock_predict_mogp(self, testing, unc=True, deriv=False, include_nugget=True, full_cov=False, **kw):
    return M.PredictResult(mean=np.vstack([mean1, mean2]), unc=np.stack([unc1] * 2) if full_cov else np.vstack([np.diag(unc1)] * 2),
                           deriv=None)


def test_reference_cases_single(monkeypatch):
    monkeypatch.setattr(M.GaussianProcessGPU, "predict", mock_predict)
    x = np.reshape(mean1, (-1, 1))
    gp = M.GaussianProcessGPU(x, targets1, nugget=0.0)
    e, P = V.standard_errors(gp, x, targets1)
    assert_allclose(e, err1[idx1] / np.sqrt(np.diag(unc1)[idx1]))
    assert np.array_equal(P, idx1)
    e, P = V.pivoted_errors(gp, x, targets1)
    A = np.linalg.cholesky(unc1[idx1][:, idx1])
    assert_allclose(e, np.linalg.solve(A, err1[idx1]), rtol=1e-12)
    assert np.array_equal(P, idx1)
    M_expect = np.dot(err1, np.linalg.solve(unc1, err1))
    assert_allclose(V.mahalanobis(gp, x, targets1), M_expect, rtol=1e-12)
    monkeypatch.setattr("scipy.stats._distn_infrastructure.rv_generic.stats", lambda *a, **k: (2.0, 3.0))
    assert_allclose(V.mahalanobis(gp, x, targets1, scaled=True), (M_expect - 2.0) / np.sqrt(3.0), rtol=1e-12)


def test_reference_cases_multi_output(monkeypatch):
    monkeypatch.setattr(M.MultiOutputGP_GPU, "predict", mock_predict_mogp)
    x = np.reshape(mean1, (-1, 1))
    T = np.vstack([targets1, targets2])
    gp = M.MultiOutputGP_GPU(x, T, nugget=0.0)
    for e, want in zip(V.standard_errors(gp, x, T), [err1[idx1] / np.sqrt(np.diag(unc1))[idx1], err2[idx1] / np.sqrt(np.diag(unc1))[idx1]]):
        assert_allclose(e[0], want)
        assert np.array_equal(e[1], idx1)
    A = np.linalg.cholesky(unc1[idx1][:, idx1])
    for e, er in zip(V.pivoted_errors(gp, x, T), [err1, err2]):
        assert_allclose(e[0], np.linalg.solve(A, er[idx1]), rtol=1e-12)
        assert np.array_equal(e[1], idx1)
    M_expect = np.array([np.dot(er, np.linalg.solve(unc1, er)) for er in (err1, err2)])
    assert_allclose(V.mahalanobis(gp, x, T), M_expect, rtol=1e-12)
    monkeypatch.setattr("scipy.stats._distn_infrastructure.rv_generic.stats", lambda *a, **k: (2.0, 3.0))
    assert_allclose(V.mahalanobis(gp, x, T, scaled=True), (M_expect - 2.0) / np.sqrt(3.0), rtol=1e-12)
