"""Two passes of the hot path (fit + gradient, predict) for rocprofv3 --pmc collection.  env: PMC_B (64), PMC_N (2000), PMC_D (10), PMC_M (2048), PMC_KERNEL (SquaredExponential)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
n, d, B, m = (int(os.environ.get(k, v)) for k, v in (("PMC_N", 2000), ("PMC_D", 10), ("PMC_B", 64), ("PMC_M", 2048)))
X, T, Xs = synth(2, n, d, B, m)
theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
gp = M.MultiOutputGP_GPU(X, T, kernel=os.environ.get("PMC_KERNEL", "SquaredExponential"), nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
for it in range(2):
    f, g, ok = gp._mogp_gpu.eval(np.tile(theta, (B, 1)), grad=True)
    mean, unc, _ = gp.predict(Xs, deriv=False)
print("ok", ok.all(), float(f.sum()))
