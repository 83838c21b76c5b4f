"""C4 (16 outputs, Matern-5/2 + fitted nugget, n=5000, d=20) and C5 (n=16000, d=8, single output):
timings + size-independent identities (no oracle at these sizes)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from bench import synth

def run(tag, cid, n, d, B, m, kernel, nugget, theta):
    X, T, Xs = synth(cid, n, d, B, m)
    nt = nugget if isinstance(nugget, str) else "fixed"
    t0 = time.perf_counter()
    gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=nugget, priors=GPPriors(n_corr=d, nugget_type=nt))
    mo = gp._mogp_gpu
    th = np.tile(theta, (B, 1))
    t1 = time.perf_counter(); f, _, ok = mo.eval(th, grad=False); t2 = time.perf_counter()
    f, _, ok = mo.eval(th, grad=False); t3 = time.perf_counter()
    f2, g, ok2 = mo.eval(th, grad=True); t4 = time.perf_counter()
    mean, unc, _ = gp.predict(Xs, deriv=False, include_nugget=False); t5 = time.perf_counter()
    mean, unc, _ = gp.predict(Xs, deriv=False, include_nugget=False); t6 = time.perf_counter()
    print("[%s] n=%d d=%d B=%d m=%d %s nugget=%s: ctor %.2fs fit(first) %.1f ms fit %.1f ms fit+grad %.1f ms predict %.1f ms (second %.1f ms) ok=%s" % (
        tag, n, d, B, m, kernel, nugget, t1 - t0, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3, ok.all() and ok2.all()))
    flops_chol = B * n ** 3 / 3.
    print("     cholesky-equivalent %.1f TF/s (fit), predict-var %.1f TF/s; |f-f2| %.2e" % (flops_chol / (t3 - t2) * 1e-12, B * m * float(n) ** 2 / (t6 - t5) * 1e-12, np.abs(f - f2).max()))
    # identities on emulator 0
    em = gp.emulators[0]
    eta = em.nugget
    a = em.Kinv_t
    tm, tv, _ = em.predict(X[:200], deriv=False, include_nugget=False)
    print("     nugget %.3e  max|mean(X_i) - (t_i - eta a_i)| = %.2e ; var range [%.2e, %.2e] (expect in [0, eta])" % (
        eta, np.abs(tm - (T[0, :200] - eta * a[:200])).max(), tv.min(), tv.max()))
    h = 1e-5
    p = 0
    e = np.zeros_like(theta); e[p] = h
    fp, _, _ = mo.eval(np.tile(theta + e, (B, 1)), grad=False); fm, _, _ = mo.eval(np.tile(theta - e, (B, 1)), grad=False)
    print("     grad[0] analytic %.8e  FD %.8e" % (g[0, p], (fp[0] - fm[0]) / (2 * h)))

only = os.environ.get("ONLY", "")
theta10 = np.array([-2. * np.log(0.3 * np.sqrt(10))] * 10 + [0.])
if only == "C2": run("C2", 2, 2000, 10, 1, 10000, "SquaredExponential", 1e-6, theta10)          # BASELINE's single-output case
if only == "S8": run("S8", 2, 2000, 10, 8, 10000, "SquaredExponential", 1e-6, theta10)          # the per-GPU shard of C3 on 8 GPUs
if only in ("C2", "S8"): sys.exit(0)
if os.environ.get("ONLY","") != "C5": run("C4", 4, 5000, 20, 16, 10000, "Matern52", "fit", np.array([-2. * np.log(0.3 * np.sqrt(20))] * 20 + [0., np.log(1e-4)]))
if os.environ.get("ONLY","") != "C4": run("C5", 5, 16000, 8, 1, 10000, "SquaredExponential", 1e-6, np.array([-2. * np.log(0.3 * np.sqrt(8))] * 8 + [0.]))
