"""Device-memory leak check of the pivoted path (per-emulator inputs, skipped-row buffers of the gradient, replica
engines of the multi-start fit)."""
import sys, gc
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import mogp_emulator_amd as M
from mogp_emulator_amd import LibGPGPU
from mogp_emulator_amd.Priors import GPPriors
rng = np.random.default_rng(0)
X0 = rng.random((500, 4)); X = np.vstack([X0, X0[[3, 99, 250]]])
T = np.stack([np.sin(X.sum(1) + k) for k in range(6)]); Xs = rng.random((300, 4))
def free():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return f / 2**20
base = None
LibGPGPU.set_fit_options(max_iter=15, ftol=1e-9, gtol=1e-6, seed=3)
for it in range(40):
    mo = M.MultiOutputGP_GPU(X, T, nugget="pivot", priors=GPPriors(n_corr=4, nugget_type="pivot"))
    th = np.tile(np.r_[np.full(4, 4.0), 0.], (6, 1))
    f, g, ok = mo._mogp_gpu.eval(th, grad=True)
    assert ok.all()
    mo.fit(th); mo.predict(Xs); mo.predict(Xs[:64], full_cov=True, deriv=False)
    if it % 4 == 0:
        M.fit_GP_MAP(mo, n_tries=2)
    LibGPGPU.pivot_cholesky(np.cov(rng.normal(size=(40, 25))) + 0.0)
    del mo; gc.collect()
    if it == 4: base = free()
    if it % 10 == 9: print(it, "free MiB", round(free()), "delta vs it=4:", round(free() - base))
