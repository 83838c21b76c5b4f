"""Wall time of fit_GP_MAP with n_tries starts on small / medium problems (the reference's own benchmark regime)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd import LibGPGPU
from bench import synth
for (B, n, d, tries) in ((1, 200, 4, 15), (8, 210, 14, 15), (32, 30, 6, 15), (16, 1000, 6, 8), (64, 2000, 10, 3)):
    X, T, _ = synth(100 + B, n, d, B, 8)
    LibGPGPU.set_fit_options(max_iter=100, ftol=1e-9, gtol=1e-6, seed=7)
    mo = M.MultiOutputGP_GPU(X, T, nugget="fit")
    t0 = time.perf_counter()
    mo = M.fit_GP_MAP(mo, n_tries=tries)
    dt = time.perf_counter() - t0
    f = np.array([em.current_logpost for em in mo.emulators])
    print("B=%3d n=%5d d=%2d n_tries=%2d: %.3f s, all fit %s, sum logpost %.6f" % (B, n, d, tries, dt, mo.get_indices_not_fit() == [], f.sum()))
