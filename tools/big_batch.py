"""Round 6: fit / fit+gradient time of LARGE batches (the replica engines of fit_GP_MAP: 96 ... 512 x n=2000) under the default schedule choice and
with the one-launch Cholesky forced (mogp_profile_schedule(4, 0)).  env: BS ("96,128,192,256"), N (2000), D (10)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd import _capi
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
lib = _capi.load()
n, d = int(os.environ.get("N", 2000)), int(os.environ.get("D", 10))
theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
for B in [int(x) for x in os.environ.get("BS", "96,128,192,256").split(",")]:
    X, T, _ = synth(2, n, d, B, 8)
    mo = M.MultiOutputGP_GPU(X, T, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))._mogp_gpu
    th = np.tile(theta, (B, 1))
    for name, sched in (("default", -1), ("one-launch", 4)):
        lib.mogp_profile_schedule(sched, 0)
        res = []
        for grad in (False, True):
            ts = []
            for it in range(6):
                t0 = time.perf_counter(); f, g, ok = mo.eval(th + 1e-3 * it, grad=grad); ts.append(time.perf_counter() - t0)
            assert ok.all()
            res.append(float(np.median(ts[2:])) * 1e3)
        print("B=%3d n=%d %-10s fit %.3f ms (%.1f us / emulator, %.1f TF)  fit+grad %.3f ms (%.1f TF)  checksum %.10g" % (
            B, n, name, res[0], res[0] / B * 1e3, B * n ** 3 / 3. / res[0] * 1e-9, res[1], B * float(n) ** 3 / res[1] * 1e-9, f.sum()), flush=True)
    lib.mogp_profile_schedule(-1, 0)
    del mo
