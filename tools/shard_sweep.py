"""Per-GPU shard regime of the C3 / C4 workloads: fit, fit+grad and predict time for
B in {8, 16, 32, 64} emulators x n=2000 x d=10 (the shards of 64 outputs over 8 / 4 / 2 / 1 GPUs) and
2 / 4 / 8 / 16 x n=5000 x d=20 Matern (C4 over 8 / 4 / 2 / 1).  One JSON line per case.
SWEEP=c3|c4|all (default c3), REPS (default 10)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
from bench import synth


def case(cid, n, d, B, m, kernel, nugget, theta, reps):
    X, T, Xs = synth(cid, n, d, B, m)
    nt = nugget if isinstance(nugget, str) else "fixed"
    gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=nugget, priors=GPPriors(n_corr=d, nugget_type=nt))
    mo = gp._mogp_gpu
    th = np.tile(theta, (B, 1))
    means = np.zeros((B, m)); vars_ = np.zeros((B, m))

    def timeit(fn):
        fn(0); fn(1)
        ts = []
        for it in range(reps):
            t0 = time.perf_counter(); fn(it + 2); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3

    t_fit = timeit(lambda it: mo.eval(th + 1e-3 * it, grad=False))
    t_fg = timeit(lambda it: mo.eval(th + 1e-3 * it, grad=True))
    t_pr = timeit(lambda it: mo.predict_variance_batch(Xs, means, vars_))
    print(json.dumps({"n": n, "d": d, "B": B, "m": m, "kernel": kernel, "fit_ms": t_fit, "fit_grad_ms": t_fg, "predict_ms": t_pr,
                      "fit_ms_per_emulator": t_fit / B, "fit_TF": B * n ** 3 / 3. / t_fit * 1e-9,
                      "fit_grad_TF": B * float(n) ** 3 / t_fg * 1e-9, "predict_TF": B * m * float(n) ** 2 / t_pr * 1e-9}), flush=True)


if __name__ == "__main__":
    which = os.environ.get("SWEEP", "c3")
    reps = int(os.environ.get("REPS", "10"))
    bs = [int(x) for x in os.environ.get("BS", "8,16,32,64").split(",")]
    if which in ("c3", "all"):
        th = np.array([-2. * np.log(0.3 * np.sqrt(10))] * 10 + [0.])
        for B in bs:
            case(2, 2000, 10, B, 10000, "SquaredExponential", 1e-6, th, reps)
    if which in ("c4", "all"):
        th = np.array([-2. * np.log(0.3 * np.sqrt(20))] * 20 + [0., np.log(1e-4)])
        for B in (2, 4, 8, 16):
            case(4, 5000, 20, B, 10000, "Matern52", "fit", th, max(3, reps // 2))
