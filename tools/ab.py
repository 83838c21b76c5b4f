"""A/B driver for library switches: every configuration (a set of MOGP_* variables) runs in its own process on the same
workload; prints fit / fit+grad / predict times and the largest differences of the results against the first configuration.

    python tools/ab.py "" "MOGP_MC_EGRP=0" "MOGP_CHOL=la"
    env: B (64), N (2000), D (10), M (10000), REPS (10), WHAT (fit,grad,predict)
"""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(out_path):
    import mogp_emulator_amd as M
    from mogp_emulator_amd.Priors import GPPriors
    from bench import synth
    B, n, d, m = (int(os.environ.get(k, v)) for k, v in (("B", 64), ("N", 2000), ("D", 10), ("M", 10000)))
    reps = int(os.environ.get("REPS", "10"))
    what = os.environ.get("WHAT", "fit,grad,predict").split(",")
    kernel = os.environ.get("KERNEL", "SquaredExponential")
    X, T, Xs = synth(2, n, d, B, m)
    gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
    mo = gp._mogp_gpu
    theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
    th = np.tile(theta, (B, 1))
    means, vars_ = np.zeros((B, m)), np.zeros((B, m))
    res = {}

    def timeit(fn):
        fn(0); fn(1)
        ts = []
        for it in range(reps):
            t0 = time.perf_counter(); fn(it + 2); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3, float(np.min(ts)) * 1e3
    if "fit" in what:
        res["fit_ms"], res["fit_min"] = timeit(lambda it: mo.eval(th + 1e-3 * it, grad=False))
    if "grad" in what:
        res["fitgrad_ms"], res["fitgrad_min"] = timeit(lambda it: mo.eval(th + 1e-3 * it, grad=True))
    f, g, ok = mo.eval(th, grad=True)
    assert ok.all()
    if "predict" in what:
        res["predict_ms"], res["predict_min"] = timeit(lambda it: mo.predict_variance_batch(Xs, means, vars_))
        res["predict_TF"] = B * m * float(n) ** 2 / res["predict_ms"] * 1e-9
    else:
        mo.predict_variance_batch(Xs[:256], means[:, :256], vars_[:, :256]) if False else None
    np.savez(out_path, f=f, g=g, mean=means, var=vars_)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    cfgs = sys.argv[1:] or [""]
    base = None
    for i, cfg in enumerate(cfgs):
        env = dict(os.environ)
        for kv in cfg.split():
            k, v = kv.split("=", 1)
            env[k] = v
        out = "/tmp/ab_%d_%d.npz" % (os.getpid(), i)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out], env=env, capture_output=True, text=True)
        if p.returncode != 0:
            print("[%s] FAILED rc=%d: %s" % (cfg, p.returncode, p.stderr[-600:]), flush=True)
            continue
        r = json.loads(p.stdout.strip().splitlines()[-1])
        z = dict(np.load(out))
        if base is None:
            base = z
        else:
            for k in ("f", "g", "mean", "var"):
                sc = max(1e-300, float(np.max(np.abs(base[k]))))
                r["d_" + k] = float(np.max(np.abs(z[k] - base[k]))) / sc
        print("[%s] %s" % (cfg, json.dumps({k: (round(v, 4) if abs(v) > 1e-3 else v) for k, v in r.items()})), flush=True)
        os.remove(out)
