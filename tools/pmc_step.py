"""One warm + one measured pass of the hot path for rocprofv3 --pmc collection (64 x n=2000 x d=10)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
n, d, B, m = 2000, 10, 64, int(os.environ.get("PMC_M", "2048"))
X, T, Xs = synth(2, n, d, B, m)
theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
gp = M.MultiOutputGP_GPU(X, T, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
for it in range(2):
    f, g, ok = gp._mogp_gpu.eval(np.tile(theta, (B, 1)), grad=True)
    mean, unc, _ = gp.predict(Xs, deriv=False)
print("ok", ok.all(), float(f.sum()))
