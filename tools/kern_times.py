"""Per-kernel HIP-event times (mogp_profile_*) of the benchmark workload for the library named by MOGP_LIB_PATH (default: the
in-tree build): fit, fit+grad and predict phases.  env: B (64), N (2000), D (10), M (10000), REPS (5), KERNEL."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd import _capi
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
lib = _capi.load()
B, n, d, m = (int(os.environ.get(k, v)) for k, v in (("B", 64), ("N", 2000), ("D", 10), ("M", 10000)))
reps = int(os.environ.get("REPS", "5"))
kernel = os.environ.get("KERNEL", "SquaredExponential")
X, T, Xs = synth(2, n, d, B, m)
gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
mo = gp._mogp_gpu
theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
th = np.tile(theta, (B, 1))
means, vars_ = np.zeros((B, m)), np.zeros((B, m))
derivs = np.zeros((B, m, d)) if os.environ.get("DERIV") else None
for it in range(2):
    mo.eval(th, grad=True); mo.eval(th, grad=False); mo.predict_variance_batch(Xs, means, vars_)
lib.mogp_profile_reset(); lib.mogp_profile_enable(1)
tf = tg = tp = 0.
for it in range(reps):
    t0 = time.perf_counter(); mo.eval(th + 1e-3 * it, grad=False); t1 = time.perf_counter()
    mo.eval(th + 1e-3 * it, grad=True); t2 = time.perf_counter()
    mo.predict_variance_batch(Xs, means, vars_); t3 = time.perf_counter()
    if os.environ.get("DERIV"):
        mo.predict_deriv(Xs, derivs)
    tf += t1 - t0; tg += t2 - t1; tp += t3 - t2
lib.mogp_profile_enable(0)
print("%s: fit %.3f ms  fit+grad %.3f ms  predict %.3f ms" % (os.environ.get("MOGP_LIB_PATH", "in-tree"), tf / reps * 1e3, tg / reps * 1e3, tp / reps * 1e3))
for tag in ("mchol", "chol_update", "chol_diag128", "chol_trsm128", "syrk_trailing", "trtri_merge", "kinv", "grad_reduce", "cov_build", "backsolve", "cross_cov", "predict_var", "predict_deriv"):
    ms, cnt, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
    if lib.mogp_profile_get(tag.encode(), ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by)) == 0 and cnt.value:
        print("   %-14s %5d launches  %9.4f ms avg  %8.2f TFLOP/s  %8.1f GB/s" % (tag, cnt.value, ms.value / cnt.value, fl.value / ms.value * 1e-9, by.value / ms.value * 1e-6))
