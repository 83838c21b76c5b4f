"""HIP-event time of the one-launch Cholesky (tag "mchol") and wall time of fit(theta) for one library configuration.
env: B (64), N (2000), D (10), REPS (10), KERNEL.  Tolerates failed factorisations (measurement switches that produce garbage)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd import _capi
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
lib = _capi.load()
B, n, d = (int(os.environ.get(k, v)) for k, v in (("B", 64), ("N", 2000), ("D", 10)))
reps = int(os.environ.get("REPS", "10"))
kernel = os.environ.get("KERNEL", "SquaredExponential")
X, T, Xs = synth(2, n, d, B, 8)
gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
mo = gp._mogp_gpu
theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
th = np.tile(theta, (B, 1))
for it in range(3):
    f, _, ok = mo.eval(th, grad=False)
lib.mogp_profile_reset(); lib.mogp_profile_enable(1)
ts = []
for it in range(reps):
    t0 = time.perf_counter(); f, _, ok = mo.eval(th + 1e-3 * it, grad=False); ts.append(time.perf_counter() - t0)
lib.mogp_profile_enable(0)
ms, cnt, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
lib.mogp_profile_get(b"mchol", ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by))
cfg = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("MOGP_"))
print("[%s] B=%d n=%d: fit %.3f ms (min %.3f)  mchol %.4f ms = %.1f TFLOP/s (%d launches)  ok=%d/%d  checksum %.12g" % (
    cfg, B, n, np.median(ts) * 1e3, min(ts) * 1e3, ms.value / max(cnt.value, 1), fl.value / max(ms.value, 1e-9) * 1e-9, cnt.value, int(np.sum(ok)), B,
    float(np.sum(f[np.isfinite(f)]))), flush=True)
