"""Round 6: tagged device times of the K build and the cross covariance (+ fit / predict wall time) for C4 and the headline shape, and the values
themselves (K entries, log-posterior) so that two builds of the library (MOGP_LIB_PATH) can be compared.  Prints one line per shape."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd import _capi
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
lib = _capi.load()

def tag_ms(tag):
    ms, cnt, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
    if lib.mogp_profile_get(tag.encode(), ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by)) == 0 and cnt.value:
        return ms.value / cnt.value
    return float("nan")

for tag, cid, n, d, B, m, kernel, nugget, theta in (
        ("C4", 4, 5000, 20, 16, 10000, "Matern52", "fit", np.array([-2. * np.log(0.3 * np.sqrt(20))] * 20 + [0., np.log(1e-4)])),
        ("C4se", 4, 5000, 20, 16, 10000, "SquaredExponential", "fit", np.array([-2. * np.log(0.3 * np.sqrt(20))] * 20 + [0., np.log(1e-4)])),
        ("C3", 2, 2000, 10, 64, 10000, "SquaredExponential", 1e-6, np.array([-2. * np.log(0.3 * np.sqrt(10))] * 10 + [0.]))):
    X, T, Xs = synth(cid, n, d, B, m)
    nt = nugget if isinstance(nugget, str) else "fixed"
    gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=nugget, priors=GPPriors(n_corr=d, nugget_type=nt))
    mo = gp._mogp_gpu
    th = np.tile(theta, (B, 1))
    means, vars_ = np.zeros((B, m)), np.zeros((B, m))
    mo.eval(th, grad=True); mo.predict_variance_batch(Xs, means, vars_)
    lib.mogp_profile_reset(); lib.mogp_profile_enable(1)
    tf, tp = [], []
    for it in range(4):
        t0 = time.perf_counter(); f, _, ok = mo.eval(th + 1e-3 * it, grad=False); tf.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); mo.predict_variance_batch(Xs, means, vars_); tp.append(time.perf_counter() - t0)
    lib.mogp_profile_enable(0)
    f, _, ok = mo.eval(th, grad=False)
    K = gp.emulators[0].get_K_matrix()
    print("%-5s lib=%s cov_build %.4f ms  cross_cov %.4f ms  fit %.3f ms  predict(host buffers) %.2f ms | sum f %.15e  K[1,0] %.17e  K.sum %.15e  mean.sum %.15e" % (
        tag, os.path.basename(os.environ.get("MOGP_LIB_PATH", "default")), tag_ms("cov_build"), tag_ms("cross_cov"), np.median(tf) * 1e3, np.median(tp) * 1e3,
        f.sum(), K[1, 0], K.sum(), means.sum()), flush=True)
    del gp, mo
