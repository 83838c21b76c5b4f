"""Parity at the FULL batches of BASELINE.json, default schedules, no environment switches:
  * C3: 64 outputs x n=2000 x d=10 on one GPU -- the two-emulator-group left-looking Cholesky (groups of 32, XCD-aware
    block decode) that bench.py times;
  * C4: 16 outputs x n=5000 x d=20, Matern-5/2, fitted nugget -- the look-ahead schedule.
Emulators compared with the oracle: first, last, the two neighbours at the stream-group boundary and others with
different XCD residues (index mod 8).  Every emulator gets its OWN theta, so a mix-up of batch slots cannot cancel.
Tolerances (the reference's own GPU-vs-CPU bar, mogp_emulator/tests/test_GaussianProcess.py:992-1071,
1098-1118): log-posterior rtol 1e-10, full gradient rtol 1e-7 (atol 1e-7 of its largest entry), 512 predictions: mean
rtol 1e-7 (atol 1e-9), variance atol 1e-7 sigma^2."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import mogp_emulator_amd as M
from oracle import cpu_ref as R
from test_gpu_parity import synth, weak

pytestmark = pytest.mark.gpu


def _ref_predict_sliced(ref, Xs, step=2500):
    """The oracle's predict over slices of X* (per-point results are independent; bounds its (n, m, d) distance temporary)."""
    parts = [ref.predict(Xs[i:i + step], include_nugget=False) for i in range(0, Xs.shape[0], step)]
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def _thetas(base, B):
    k = np.arange(B)[:, None]
    return base[None, :] + 0.05 * np.sin(1.0 + k + 0.37 * np.arange(base.size)[None, :])


def test_c3_full_batch_64_default_schedule_vs_oracle():
    n, d, B, m = 2000, 10, 64, 512
    X, T, Xs = synth(20240607 + 3, n, d, B, m)
    eta = 1e-6
    th = _thetas(np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.]), B)
    mo = M.MultiOutputGP_GPU(X, T, nugget=eta, priors=weak(d, eta))
    f, g, ok = mo._mogp_gpu.eval(th, grad=True)
    assert ok.all() and np.all(np.isfinite(g))
    f_only, _, ok2 = mo._mogp_gpu.eval(th, grad=False)          # the objective-only path (alpha by back substitution)
    assert ok2.all()
    mo.fit(th)
    mean, unc, _ = mo.predict(Xs, deriv=False)
    # first / last, both sides of the stream-group boundary (31 | 32), and residues 13 % 8 = 5, 42 % 8 = 2, 54 % 8 = 6
    for k in (0, 63, 31, 32, 13, 42, 54):
        ref = R.GPRef(X, T[k], nugget=eta)
        lp = ref.fit(th[k])
        assert_allclose(f[k], lp, rtol=1e-10, err_msg="emulator %d" % k)
        assert_allclose(f_only[k], lp, rtol=1e-10, err_msg="emulator %d" % k)
        gref = ref.logpost_deriv(th[k])
        assert_allclose(g[k], gref, rtol=1e-7, atol=1e-7 * np.abs(gref).max(), err_msg="emulator %d" % k)
        mu, var, _ = ref.predict(Xs)
        assert_allclose(mean[k], mu, rtol=1e-7, atol=1e-9, err_msg="emulator %d" % k)
        assert_allclose(unc[k], var, atol=1e-7, err_msg="emulator %d" % k)
        assert_allclose(mo.emulators[k].Kinv_t, ref.Kinv_t, rtol=1e-6, atol=1e-7 * np.abs(ref.Kinv_t).max())
    # every emulator of the batch: mean at training points = t - eta alpha (an identity that needs no oracle)
    tm, _, _ = mo.predict(X[:128], deriv=False, include_nugget=False)
    for k in range(B):
        a = mo.emulators[k].Kinv_t
        assert_allclose(tm[k], T[k, :128] - eta * a[:128], rtol=1e-7, atol=1e-8, err_msg="emulator %d" % k)


def test_c4_full_batch_16_default_schedule_vs_oracle():
    n, d, B, m = 5000, 20, 16, 512
    X, T, Xs = synth(4, n, d, B, m)
    th = _thetas(np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0., np.log(1e-4)]), B)
    mo = M.MultiOutputGP_GPU(X, T, kernel="Matern52", nugget="fit", priors=weak(d, "fit"))
    f, g, ok = mo._mogp_gpu.eval(th, grad=True)
    assert ok.all() and np.all(np.isfinite(g))
    mo.fit(th)
    mean, unc, _ = mo.predict(Xs, deriv=False)
    for j, k in enumerate((0, 15, 7, 10)):                     # residues 0, 7, 7, 2; first / last / middle
        ref = R.GPRef(X, T[k], kernel="Matern52", nugget="fit", chunk_rows=256)
        assert_allclose(f[k], ref.fit(th[k]), rtol=1e-10, err_msg="emulator %d" % k)
        mu, var, _ = ref.predict(Xs)
        assert_allclose(mean[k], mu, rtol=1e-7, atol=1e-9, err_msg="emulator %d" % k)
        assert_allclose(unc[k], var, atol=1e-7, err_msg="emulator %d" % k)
        assert_allclose(mo.emulators[k].nugget, np.exp(th[k, -1]), rtol=1e-14)
        if j < 2:                                              # full gradient (22 components) of the first and the last
            gref = ref.logpost_deriv_chunked(th[k], chunk_rows=256)
            assert_allclose(g[k], gref, rtol=1e-7, atol=1e-7 * np.abs(gref).max(), err_msg="emulator %d" % k)
    eta = np.exp(th[:, -1])
    tm, tv, _ = mo.predict(X[:128], deriv=False, include_nugget=False)
    for k in range(B):
        a = mo.emulators[k].Kinv_t
        assert_allclose(tm[k], T[k, :128] - eta[k] * a[:128], rtol=1e-8, atol=1e-9, err_msg="emulator %d" % k)
        assert np.all(tv[k] >= 0.) and np.all(tv[k] <= eta[k] * (1 + 1e-9))


def test_c2_single_output_fit_and_predict_at_all_10k_points_vs_oracle():
    """BASELINE's C2 through the wrapper every user of the reference calls (GaussianProcessGPU.fit / .predict,
    GaussianProcessGPU.py:431-438, 560-626): one n=2000 emulator, the chain-bound single-matrix Cholesky, and the prediction at
    ALL m = 10 000 points in one call -- mean rtol 1e-7, variance atol 1e-7 (tests/test_GaussianProcess.py:992-1071)."""
    n, d, m = 2000, 10, 10000
    X, T, Xs = synth(20240607 + 2, n, d, 1, m)
    eta = 1e-6
    theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
    gp = M.GaussianProcessGPU(X, T[0], nugget=eta, priors=weak(d, eta), max_batch_size=m)
    gp.fit(theta)
    ref = R.GPRef(X, T[0], nugget=eta)
    assert_allclose(gp.current_logpost, ref.fit(theta), rtol=1e-10)
    gref = ref.logpost_deriv(theta)
    assert_allclose(gp.logpost_deriv(theta), gref, rtol=1e-7, atol=1e-7 * np.abs(gref).max())
    mean, unc, deriv = gp.predict(Xs, include_nugget=False)
    assert mean.shape == (m,) and unc.shape == (m,) and deriv.shape == (m, d)
    mu, var = _ref_predict_sliced(ref, Xs)
    assert_allclose(mean, mu, rtol=1e-7, atol=1e-9)
    assert_allclose(unc, var, atol=1e-7)
    _, _, dref = ref.predict(Xs[:200], deriv=True)
    assert_allclose(deriv[:200], dref, rtol=1e-7, atol=1e-8)


def test_c3_device_resident_predict_of_10k_points_in_one_launch_vs_oracle():
    """The predict phase bench.py times: 64 emulators x m = 10 000 points through predict_variance_batch_dev (X* and the outputs
    in HBM, ONE cross-covariance chunk of 64 x 10112 x 2048 doubles, one predictive-variance launch), two emulators checked against
    the oracle at every point and all 64 against the host-buffer entry point."""
    import torch
    n, d, B, m = 2000, 10, 64, 10000
    X, T, Xs = synth(2, n, d, B, m)
    eta = 1e-6
    th = _thetas(np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.]), B)
    mo = M.MultiOutputGP_GPU(X, T, nugget=eta, priors=weak(d, eta))
    mo.fit(th)
    dev = torch.device("cuda", 0)
    d_Xs = torch.from_numpy(Xs).to(dev)
    d_mean = torch.full((B, m), float("nan"), dtype=torch.float64, device=dev)
    d_var = torch.full((B, m), float("nan"), dtype=torch.float64, device=dev)
    mo._mogp_gpu.predict_variance_batch_dev(d_Xs.data_ptr(), m, d_mean.data_ptr(), d_var.data_ptr())
    torch.cuda.synchronize()
    mean, var = d_mean.cpu().numpy(), d_var.cpu().numpy()
    assert np.all(np.isfinite(mean)) and np.all(np.isfinite(var))
    for k in (5, 58):
        ref = R.GPRef(X, T[k], nugget=eta)
        ref.fit(th[k])
        mu, v = _ref_predict_sliced(ref, Xs)
        assert_allclose(mean[k], mu, rtol=1e-7, atol=1e-9, err_msg="emulator %d" % k)
        assert_allclose(np.maximum(var[k], 0.), v, atol=1e-7, err_msg="emulator %d" % k)
    hm, hv = np.zeros((B, m)), np.zeros((B, m))
    mo._mogp_gpu.predict_variance_batch(Xs, hm, hv)
    assert_allclose(mean, hm, rtol=1e-12, atol=1e-12)
    assert_allclose(var, hv, rtol=0, atol=1e-11)
