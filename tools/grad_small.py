"""objective + gradient evaluations of B x n=2000 x d=10 (B from env, default 64) for timeline analysis under rocprofv3."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
n, d, B = 2000, 10, int(os.environ.get("B", "64"))
X, T, Xs = synth(2, n, d, B, 8)
theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
gp = M.MultiOutputGP_GPU(X, T, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
th = np.tile(theta, (B, 1))
for it in range(3):
    gp._mogp_gpu.eval(th, grad=True)
t0 = time.perf_counter()
for it in range(20):
    gp._mogp_gpu.eval(th + 1e-3 * it, grad=True)
print("ms per fit+grad", (time.perf_counter() - t0) / 20 * 1e3)
