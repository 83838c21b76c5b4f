"""Shapes far from the benchmark configurations: many small emulators, a long prediction sweep, n just above tile edges."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from oracle import cpu_ref as R
from bench import synth

# 1. many small emulators
B, n, d, m = 500, 200, 4, 300
X, T, Xs = synth(71, n, d, B, m)
theta = np.r_[np.full(d, -2 * np.log(0.3 * np.sqrt(d))), 0.2]
mo = M.MultiOutputGP_GPU(X, T, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
t0 = time.perf_counter(); f, g, ok = mo._mogp_gpu.eval(np.tile(theta, (B, 1)), grad=True); t1 = time.perf_counter()
mo.fit(np.tile(theta, (B, 1)))
mean, unc, _ = mo.predict(Xs, deriv=False); t2 = time.perf_counter()
err = 0.
for k in (0, 137, 499):
    ref = R.GPRef(X, T[k], nugget=1e-6); lp = ref.fit(theta)
    err = max(err, abs(f[k] - lp) / abs(lp), np.abs(g[k] - ref.logpost_deriv(theta)).max() / np.abs(g[k]).max(), np.abs(mean[k] - ref.predict(Xs)[0]).max())
print("500 x n=200: fit+grad %.1f ms, predict %.1f ms, ok=%s, max err vs oracle %.2e" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, ok.all(), err))

# 2. long sweep on few emulators (several device chunks) + implausibility
B, n, d, m = 3, 1500, 6, 400000
X, T, Xs = synth(72, n, d, B, m)
theta = np.r_[np.full(d, -2 * np.log(0.3 * np.sqrt(d))), 0.0]
mo = M.MultiOutputGP_GPU(X, T, nugget=1e-5, priors=GPPriors(n_corr=d, nugget_type="fixed"))
mo.fit(np.tile(theta, (B, 1)))
t0 = time.perf_counter(); mean, unc, _ = mo.predict(Xs, deriv=False); t1 = time.perf_counter()
I = mo._mogp_gpu.implausibility(Xs, np.zeros(B), np.full(B, 0.01), np.zeros(B), rank=1); t2 = time.perf_counter()
ref = R.GPRef(X, T[1], nugget=1e-5); ref.fit(theta)
sel = np.r_[0:50, m - 50:m]
rmu, rvar, _ = ref.predict(Xs[sel])
print("3 x n=1500 x m=4e5: predict %.0f ms (%.1f M pts/s), implausibility %.0f ms, mean err %.2e var err %.2e, I vs host %.2e" % (
    (t1 - t0) * 1e3, B * m / (t1 - t0) / 1e6, (t2 - t1) * 1e3, np.abs(mean[1][sel] - rmu).max(), np.abs(unc[1][sel] - rvar).max(),
    np.abs(I - R.implausibility_ref(np.zeros(B), np.full(B, 0.01), mean, unc, 0., 1)).max()))

# 3. single emulator, n just past a tile edge, Matern + fitted nugget
n, d, m = 4097, 5, 1000
X, T, Xs = synth(73, n, d, 1, m)
theta = np.r_[np.full(d, -2 * np.log(0.3 * np.sqrt(d))), 0.0, np.log(1e-4)]
gp = M.GaussianProcessGPU(X, T[0], kernel="Matern52", nugget="fit", priors=GPPriors(n_corr=d, nugget_type="fit"))
t0 = time.perf_counter(); lp = gp.logposterior(theta); g = gp.logpost_deriv(theta); t1 = time.perf_counter()
ref = R.GPRef(X, T[0], kernel="Matern52", nugget="fit", chunk_rows=512); rl = ref.fit(theta)
mu, var, _ = gp.predict(Xs, deriv=False)
rmu, rvar, _ = ref.predict(Xs)
print("n=4097 Matern fit-nugget: %.0f ms, logpost rel err %.2e, mean err %.2e, var err %.2e" % ((t1 - t0) * 1e3, abs(lp - rl) / abs(rl), np.abs(mu - rmu).max(), np.abs(var - rvar).max()))
