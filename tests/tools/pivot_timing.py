"""Wall time of fit(theta) with nugget="pivot" (BLAS-2 pivoted Cholesky, one workgroup per emulator) next to the blocked
MFMA path (nugget fixed) on the device and LAPACK dpstrf on the host (oracle, one emulator)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
from oracle import cpu_ref as R
for (B, n, d) in ((8, 210, 14), (64, 500, 10), (64, 1000, 10), (64, 2000, 10), (256, 2000, 10), (4, 5000, 10)):
    X, T, _ = synth(300 + n, n, d, B, 8)
    corr = -2 * np.log(0.3 * np.sqrt(d))
    theta = np.tile(np.array([corr] * d + [0.0]), (B, 1))
    res = {}
    for tag, nug in (("pivot", "pivot"), ("fixed", 1e-6)):
        mo = M.MultiOutputGP_GPU(X, T, nugget=nug, priors=GPPriors(n_corr=d, nugget_type="pivot" if tag == "pivot" else "fixed"))
        mo.fit(theta)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            mo.fit(theta)
        res[tag] = (time.perf_counter() - t0) / reps
    ref = R.GPRef(X, T[0], nugget="pivot", chunk_rows=512)
    t0 = time.perf_counter()
    K = ref.get_cov_matrix(X) if False else None
    ref.theta = theta[0]
    Kmat = np.exp(theta[0][d]) * R.calc_K(R.calc_r2_chunked(X, X, theta[0][:d], 512))
    t1 = time.perf_counter()
    R.pivot_cholesky(Kmat)
    cpu = time.perf_counter() - t1
    print("B=%3d n=%5d: device pivot %.4f s (%.1f fits/s), device blocked %.4f s, host dpstrf %.3f s per emulator"
          % (B, n, res["pivot"], B / res["pivot"], res["fixed"], cpu), flush=True)
