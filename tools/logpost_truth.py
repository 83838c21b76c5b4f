"""How far is the log-posterior of an ill-conditioned fit from its exact value?  The oracle (LAPACK), and -- with a GPU -- the library,
against a Cholesky of the same fp64 matrix in 80-bit long double (numpy longdouble, blocked, no BLAS).  Answers whether a difference
of a few 1e-10 between two factorisations is error or noise:   python tools/logpost_truth.py [truth|gpu]
env: N (1500), D (6), B (5), SEED (78), NUGGET (1e-6)"""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
n, d, B, seed = (int(os.environ.get(k, v)) for k, v in (("N", 1500), ("D", 6), ("B", 5), ("SEED", 78)))
nug = float(os.environ.get("NUGGET", "1e-6"))


def synth(seed, n, d, n_out, m):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (n, d))
    T = np.empty((n_out, n))
    for k in range(n_out):
        w = rng.normal(size=d)
        T[k] = np.sin(2 * np.pi * X @ w / np.sqrt(d)) + 0.1 * (X ** 2) @ np.abs(w) + 0.01 * rng.normal(size=n)
    return X, T, rng.uniform(0, 1, (m, d))


X, T, _ = synth(seed, n, d, B, 50)
theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
what = sys.argv[1] if len(sys.argv) > 1 else "truth"
if what == "truth":
    from oracle import cpu_ref as R
    ref = R.GPRef(X, T[0], nugget=nug)
    ref.fit(theta)
    K = ref.get_K_matrix().copy()
    K[np.diag_indices(n)] += nug
    from oracle.exact import loglike_longdouble, cond_eps
    print("cond(K) eps = %.3e" % cond_eps(K))
    t0 = time.time()
    like = loglike_longdouble(K, T)
    print("long double Cholesky + solves: %.1f s" % (time.time() - t0))
    # the prior term is theta-only: take it from the oracle as (its f) - (its likelihood part)
    for k in range(B):
        r = R.GPRef(X, T[k], nugget=nug); fk = r.fit(theta)
        prior = fk - 0.5 * (np.dot(r.t, r.Kinv_t) + R.logdet_L(r.L) + n * np.log(2. * np.pi))
        ft = like[k] + np.longdouble(prior)
        print("k=%d  exact %s   oracle %.17g   (oracle - exact) / exact = %.3e" % (k, np.format_float_positional(ft, precision=22), fk, float((np.longdouble(fk) - ft) / ft)))
else:
    import mogp_emulator_amd as M
    from mogp_emulator_amd.Priors import GPPriors
    mo = M.MultiOutputGP_GPU(X, T, nugget=nug, priors=GPPriors(n_corr=d, nugget_type="fixed"))
    f, g, ok = mo._mogp_gpu.eval(np.tile(theta, (B, 1)), grad=False)
    print(os.environ.get("MOGP_LIB_PATH", "in-tree"), os.environ.get("MOGP_CHOL", ""), " ".join("%.17g" % x for x in f))
