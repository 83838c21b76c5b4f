import sys, gc, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
rng = np.random.default_rng(0)
X = rng.random((700, 4)); T = np.stack([np.sin(X.sum(1) + k) for k in range(6)]); Xs = rng.random((300, 4))
def free():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return f / 2**20
base = None
for it in range(60):
    mo = M.MultiOutputGP_GPU(X, T, nugget="fit", priors=GPPriors(n_corr=4, nugget_type="fit"), analytic_mean=(it % 2 == 0), mean=M.LibGPGPU.ConstMeanFunc() if it % 2 == 0 else None)
    th = np.tile(np.r_[np.ones(4), 0., -8.], (6, 1))
    mo._mogp_gpu.eval(th, grad=True); mo.fit(th)
    mo.predict(Xs); mo.predict(Xs[:64], full_cov=True, deriv=False)
    if it % 2 == 1:
        mo._mogp_gpu.implausibility(Xs, np.zeros(6), np.ones(6) * .01, np.zeros(6), rank=1)
    del mo; gc.collect()
    if it == 4: base = free()
    if it % 10 == 9: print(it, "free MiB", round(free()), "delta vs it=4:", round(free() - base))
