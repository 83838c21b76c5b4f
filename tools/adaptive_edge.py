"""Round 6: what does the device decide at the adaptive-nugget knife-edge (zero jitter on a cond ~ 1/eps matrix,
mogp_emulator/linalg/cholesky.py:234-281)?  Prints device vs golden for the c1_n200_d4 / n500_d10 *_adaptive_* fixtures, alone and inside
batches, so that the bars of tests/test_gpu_parity.py::test_medium_configs_vs_reference[adaptive] can be stated from measurements."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from conftest import load_golden

def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

for tag in ("c1_n200_d4", "n500_d10"):
    g = load_golden(tag + ".npz")
    X = g["X"]; D = X.shape[1]
    for kern in ("SquaredExponential", "Matern52"):
        pre = "%s_adaptive_" % kern
        theta = g[pre + "theta"]
        gp = M.GaussianProcessGPU(X, g["T"][0], kernel=kern, nugget="adaptive", priors=GPPriors(n_corr=D, nugget_type="adaptive"))
        gp.fit(theta)
        K = gp.get_K_matrix()
        ev = np.linalg.eigvalsh(K)
        mean, unc, _ = gp.predict(g["Xs"])
        grad = gp.logpost_deriv(theta)
        print("%s %s: nugget dev %r gold %r | cond %.2e | logpost dev %.10f gold %.10f rel %.2e | Ldiag rel %.2e min Ldiag dev %.3e gold %.3e | alpha rel %.2e | grad rel %.2e | mean rel %.2e | var absmax %.2e (gold max %.2e)"
              % (tag, kern, gp.nugget, float(g[pre + "nugget"]), ev[-1] / max(ev[0], 1e-300), gp.current_logpost, float(g[pre + "logpost"]),
                 rel(gp.current_logpost, g[pre + "logpost"]), rel(np.diag(gp.L), g[pre + "L_diag"]), np.diag(gp.L).min(), g[pre + "L_diag"].min(),
                 rel(gp.Kinv_t, g[pre + "alpha"]), rel(grad, g[pre + "grad"]), rel(mean, g[pre + "mean"]), np.abs(unc - g[pre + "var"]).max(), np.abs(g[pre + "var"]).max()), flush=True)
        for B in (1, 3, 8, 16):
            T = np.tile(g["T"][0], (B, 1))
            mo = M.MultiOutputGP_GPU(X, T, kernel=kern, nugget="adaptive", priors=GPPriors(n_corr=D, nugget_type="adaptive"))
            mo.fit(np.tile(theta, (B, 1)))
            nug = mo._nuggets()
            lp = np.array([e.current_logpost for e in mo.emulators])
            print("   batch %2d (MOGP_CHOL=%s): nuggets %s  logpost spread %.2e  vs solo %.2e" % (B, os.environ.get("MOGP_CHOL", "default"), sorted(set(nug.tolist())), np.ptp(lp), rel(lp[0], gp.current_logpost)), flush=True)
