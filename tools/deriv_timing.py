"""predict_deriv timing (64 x n=2000 x d=10, m=10^4 by default; env B N D M KERNEL): device time through the host-buffer API."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
B, n, d, m = (int(os.environ.get(k, v)) for k, v in (("B", 64), ("N", 2000), ("D", 10), ("M", 10000)))
kernel = os.environ.get("KERNEL", "SquaredExponential")
X, T, Xs = synth(2, n, d, B, m)
gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
gp.fit(np.tile(theta, (B, 1)))
out = np.zeros((B, m, d))
gp._mogp_gpu.predict_deriv(Xs, out)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); gp._mogp_gpu.predict_deriv(Xs, out); ts.append(time.perf_counter() - t0)
print("%s B=%d n=%d d=%d m=%d %s: predict_deriv %.2f ms (host buffers, D2H of %.0f MB included)  checksum %.12e" % (
    os.environ.get("MOGP_LIB_PATH", "in-tree"), B, n, d, m, kernel, min(ts) * 1e3, out.nbytes / 1e6, float(np.abs(out).sum())))
