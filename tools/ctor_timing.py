import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from mogp_emulator_amd import _capi
from bench import synth
lib = _capi.load()
n, d = 2000, 10
theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
for B in (64, 256, 512, 960, 512, 256):
    X, T, Xs = synth(2, n, d, B, 8)
    t0 = time.perf_counter()
    gp = M.MultiOutputGP_GPU(X, T, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
    lib.mogp_dev_synchronize()
    t1 = time.perf_counter()
    th = np.tile(theta, (B, 1))
    ts = []
    for it in range(4):
        a = time.perf_counter(); gp._mogp_gpu.eval(th + 1e-3 * it, grad=True); ts.append(time.perf_counter() - a)
    t2 = time.perf_counter()
    del gp
    lib.mogp_dev_synchronize()
    t3 = time.perf_counter()
    print("B=%d ctor %.3f s, evals %s ms, dtor %.3f s" % (B, t1 - t0, " ".join("%.1f" % (x * 1e3) for x in ts), t3 - t2), flush=True)
