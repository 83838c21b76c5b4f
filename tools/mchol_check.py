"""One-launch Cholesky (schedule 4, kernels_mchol.hip) against the library's default schedule: results and fit time.
CONFIGS env: comma list of B:n:d[:m] (m = Matern + fitted nugget)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd import _capi
from mogp_emulator_amd.Priors import GPPriors
from bench import synth

lib = _capi.load()


def counter(name):
    v = ctypes.c_longlong()
    lib.mogp_profile_counter(name.encode(), ctypes.byref(v))
    return int(v.value)


def run(B, n, d, matern, reps):
    X, T, _ = synth(7, n, d, B, 8)
    if matern:
        kernel, nugget = "Matern52", "fit"
        theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0., np.log(1e-4)])
    else:
        kernel, nugget = "SquaredExponential", 1e-6
        theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
    nt = nugget if isinstance(nugget, str) else "fixed"
    gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=nugget, priors=GPPriors(n_corr=d, nugget_type=nt))
    mo = gp._mogp_gpu
    th = np.tile(theta, (B, 1)) + 0.01 * np.sin(np.arange(B))[:, None]
    res = {}
    for name, sched in (("default", int(os.environ.get("BASE_SCHED", "-1"))), ("mchol", 4)):
        lib.mogp_profile_schedule(sched, 0)
        f, _, ok = mo.eval(th, grad=False)
        f, _, ok = mo.eval(th, grad=False)
        ts = []
        for it in range(reps):
            t0 = time.perf_counter(); mo.eval(th + 1e-4 * (it + 1), grad=False); ts.append(time.perf_counter() - t0)
        f, g, ok2 = mo.eval(th, grad=True)
        tg = []
        for it in range(max(2, reps // 2)):
            t0 = time.perf_counter(); mo.eval(th + 1e-4 * (it + 1), grad=True); tg.append(time.perf_counter() - t0)
        f, g, ok2 = mo.eval(th, grad=True)
        gp.fit(th)
        a = np.stack([gp.emulators[k].Kinv_t for k in sorted(set([0, B - 1]))])
        L = gp.emulators[B - 1].L if n <= 2000 else None
        res[name] = (f.copy(), g.copy(), a, L, float(np.median(ts)) * 1e3, float(np.median(tg)) * 1e3, bool(ok.all() and ok2.all()))
    lib.mogp_profile_schedule(-1, 0)
    f0, g0, a0, L0, t0, tg0, ok0 = res["default"]
    f1, g1, a1, L1, t1, tg1, ok1 = res["mchol"]
    df = np.max(np.abs(f1 - f0) / np.abs(f0))
    dg = np.max(np.abs(g1 - g0)) / np.max(np.abs(g0))
    da = np.max(np.abs(a1 - a0)) / np.max(np.abs(a0))
    dL = -1.0 if L0 is None else np.max(np.abs(L1 - L0)) / np.max(np.abs(L0))
    print("B=%3d n=%5d d=%2d %-8s fit default %8.3f ms  mchol %8.3f ms (x%.2f)   fit+grad %8.3f -> %8.3f   rel diff logpost %.1e grad %.1e alpha %.1e L %.1e  ok %s/%s aborts %d" % (
        B, n, d, "matern" if matern else "sqexp", t0, t1, t0 / t1, tg0, tg1, df, dg, da, dL, ok0, ok1, counter("mchol_aborts")), flush=True)


cfgs = os.environ.get("CONFIGS", "1:100:3,5:130:4,3:700:5,8:2000:10,16:2000:10,32:2000:10,64:2000:10,2:5000:20:m,16:5000:20:m,1:16000:8")
for c in cfgs.split(","):
    p = c.split(":")
    run(int(p[0]), int(p[1]), int(p[2]), len(p) > 3, int(os.environ.get("REPS", "7")))
