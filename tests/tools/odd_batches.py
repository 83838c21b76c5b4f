import sys, numpy as np
sys.path.insert(0, "/root/repo")
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from oracle import cpu_ref as R
from bench import synth
for B, n in ((17, 300), (23, 513), (31, 129), (33, 1000), (16, 64), (19, 2000)):
    X, T, Xs = synth(B, n, 3, B, 40)
    rng = np.random.default_rng(B)
    th = np.stack([np.r_[rng.uniform(0, 2, 3), rng.uniform(-1, 1)] for _ in range(B)])
    mo = M.MultiOutputGP_GPU(X, T, nugget=1e-5, priors=GPPriors(n_corr=3, nugget_type="fixed"))
    f, g, ok = mo._mogp_gpu.eval(th, grad=True)
    mo.fit(th)
    mean, unc, _ = mo.predict(Xs, deriv=False)
    err = 0.
    for k in (0, B // 2, B - 1):
        ref = R.GPRef(X, T[k], nugget=1e-5); lp = ref.fit(th[k])
        rm, rv, _ = ref.predict(Xs)
        err = max(err, abs(f[k] - lp) / abs(lp), np.abs(g[k] - ref.logpost_deriv(th[k])).max() / np.abs(g[k]).max(), np.abs(mean[k] - rm).max(), np.abs(unc[k] - rv).max())
    print(B, n, ok.all(), "max err %.2e" % err)
