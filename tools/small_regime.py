"""Many small emulators (the regime of the reference's own benchmarks): wall time per batched objective / objective+gradient evaluation."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
for (B, n, d) in ((500, 200, 4), (64, 200, 4), (8, 210, 14), (2000, 100, 3)):
    X, T, Xs = synth(71, n, d, B, 10)
    theta = np.r_[np.full(d, -2 * np.log(0.3 * np.sqrt(d))), 0.2]
    mo = M.MultiOutputGP_GPU(X, T, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
    th = np.tile(theta, (B, 1))
    mo._mogp_gpu.eval(th, grad=True)
    t0 = time.perf_counter()
    for i in range(10): mo._mogp_gpu.eval(th + 1e-3 * i, grad=False)
    t1 = time.perf_counter()
    for i in range(10): mo._mogp_gpu.eval(th + 1e-3 * i, grad=True)
    t2 = time.perf_counter()
    print("B=%d n=%d: fit %.2f ms, fit+grad %.2f ms" % (B, n, (t1 - t0) * 100, (t2 - t1) * 100))
