"""
Generate the golden vectors under tests/golden/ by IMPORTING THE REAL REFERENCE.

Runs only in the build container (never on the GPU box, never from tests):

    mkdir -p /tmp/ref_oracle && cp -r /root/reference/mogp_emulator /tmp/ref_oracle/
    echo "version = '0.7.2'" > /tmp/ref_oracle/mogp_emulator/version.py   # setup.py:23-34 generates this
    cd /root/repo && PYTHONPATH=/tmp/ref_oracle /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden.py

(/opt/conda/bin/python3.9 has patsy, which mogp_emulator/GaussianProcess.py:10-13
hard-requires; the system python3 does not.)

The outputs are data only: inputs + the reference's outputs on them.
"""
import os
import sys
import numpy as np

import mogp_emulator
from mogp_emulator import GaussianProcess, MultiOutputGP, fit_GP_MAP
from mogp_emulator.Kernel import SquaredExponential, Matern52
from mogp_emulator.Priors import (GPPriors, InvGammaPrior, GammaPrior, LogNormalPrior, WeakPrior,
                                  min_spacing, max_spacing)
from mogp_emulator.GPParams import GPParams
from mogp_emulator.linalg.cholesky import jit_cholesky, fixed_cholesky

HERE = os.path.dirname(os.path.abspath(__file__))
KERNELS = {"SquaredExponential": SquaredExponential, "Matern52": Matern52}


def synth(config_id, n, d, n_out, m):
    """SURVEY.md section 8d synthetic generator."""
    rng = np.random.default_rng(20240607 + config_id)
    X = rng.uniform(0, 1, (n, d))
    T = np.empty((n_out, n))
    for k in range(n_out):
        w = rng.normal(size=d)
        T[k] = np.sin(2 * np.pi * X @ w / np.sqrt(d)) + 0.1 * (X ** 2) @ np.abs(w) + 0.01 * rng.normal(size=n)
    Xs = rng.uniform(0, 1, (m, d))
    return X, T, Xs


def weak(n_corr, nugget_type):
    return GPPriors(n_corr=n_corr, nugget_type=nugget_type)


def run_gp(X, t, kernel, nugget, theta, Xs, priors="weak"):
    nugget_type = nugget if isinstance(nugget, str) else "fixed"
    pri = weak(X.shape[1], nugget_type) if priors == "weak" else None
    gp = GaussianProcess(X, t, kernel=KERNELS[kernel](), nugget=nugget, priors=pri)
    gp.fit(np.array(theta))
    out = dict(logpost=gp.current_logpost, nugget=gp.nugget, L=gp.Kinv.L, alpha=gp.Kinv_t,
               grad=gp.logpost_deriv(np.array(theta)), K=gp.get_K_matrix())
    if Xs is not None:
        mean, var, _ = gp.predict(Xs)
        mean2, var_nonug, _ = gp.predict(Xs, include_nugget=False)
        out.update(mean=mean, var=var, var_nonug=var_nonug)
    return gp, out


def pivot_section():
    """---- 15. nugget="pivot": pivoted Cholesky (linalg/cholesky.py:82-165, 284-327), SURVEY 8f row 4 ----"""
    from mogp_emulator.linalg.cholesky import pivot_cholesky
    out = {}
    # the two matrices of tests/test_linalg.py:156-188 and a larger rank-deficient one
    rng = np.random.default_rng(1515)
    G = rng.normal(size=(12, 7))
    mats = {"wiki": np.array([[4., 12., -16.], [12., 37., -43.], [-16., -43., 98.]]),
            "collinear": np.array([[1., 1., 1.e-6], [1., 1., 1.e-6], [1.e-6, 1.e-6, 1.]]),
            "gram_rank7": G @ G.T + 0.0}
    for tag, A in mats.items():
        L, P = pivot_cholesky(np.copy(A))
        out["mat_%s_A" % tag], out["mat_%s_L" % tag], out["mat_%s_P" % tag] = A, L, np.asarray(P, dtype=np.int64)
    # the three-point emulator of tests/test_GaussianProcess.py:397-415, 1120-1143
    x2, y2 = np.array([1., 2., 4.]), np.array([1., 2., 1.])
    gp = GaussianProcess(x2, y2, nugget="pivot")
    gp.theta = np.zeros(2)
    xpred = np.linspace(0., 5.)
    mean, var, _ = gp.predict(xpred)
    out.update(three_x=x2, three_y=y2, three_xpred=xpred, three_L=gp.Kinv.L, three_P=np.asarray(gp.Kinv.P, dtype=np.int64),
               three_Kinv_t=gp.Kinv_t, three_mean=mean, three_var=var, three_logpost=np.array(gp.current_logpost))
    # emulators: full rank, and with repeated design points (rank-deficient K)
    X, T, Xs = synth(15, 40, 3, 1, 30)
    t = T[0] + 0.5 + 1.2 * X[:, 0]
    Xd = np.vstack([X[:7], X[3:4], X[7:25], X[11:12], X[20:21], X[25:]])          # rows 3, 11, 20 repeated
    td_same = np.concatenate([t[:7], t[3:4], t[7:25], t[11:12], t[20:21], t[25:]])  # ... with the same targets
    td_diff = td_same.copy()
    td_diff[[7, 26, 27]] += np.array([0.05, -0.03, 0.02])                           # ... with different targets
    out.update(X=X, t=t, Xs=Xs, Xd=Xd, td_same=td_same, td_diff=td_diff)
    for tag, (XX, tt) in {"full": (X, t), "dupsame": (Xd, td_same), "dupdiff": (Xd, td_diff)}.items():
        for kern in KERNELS:
            for mtag, formula in (("zero", None), ("lin", "x[0]")):
                theta = np.array([3.0, 2.5, 3.5, 0.2])      # short length scales: only the repeats make K singular
                gp = GaussianProcess(XX, tt, mean=formula, kernel=KERNELS[kern](), nugget="pivot", priors=weak(3, "pivot"))
                gp.fit(theta)
                pre = "%s_%s_%s_" % (tag, kern, mtag)
                out[pre + "theta"] = theta
                out[pre + "logpost"] = np.array(gp.current_logpost)
                out[pre + "grad"] = gp.logpost_deriv(theta)
                out[pre + "L"] = gp.Kinv.L
                out[pre + "P"] = np.asarray(gp.Kinv.P, dtype=np.int64)
                out[pre + "Kinv_t"] = gp.Kinv_t
                out[pre + "Kinv_t_mean"] = gp.Kinv_t_mean
                out[pre + "beta"] = np.array(gp.theta.mean)
                mean, var, deriv = gp.predict(Xs)
                out[pre + "mean"], out[pre + "var"] = mean, var
                out[pre + "var_nonug"] = gp.predict(Xs, include_nugget=False)[1]
                out[pre + "cov"] = gp.predict(Xs, full_cov=True)[1]
                out[pre + "nugget_is_none"] = np.array(gp.nugget is None)
    # fit_GP_MAP with pivoting on the full-rank set (end point only: optimiser trajectories are not pinned)
    np.random.seed(1515)
    gp = GaussianProcess(X, t, nugget="pivot")
    gp = fit_GP_MAP(gp, n_tries=3)
    out["map_theta"] = gp.theta.get_data()
    out["map_logpost"] = np.array(gp.current_logpost)
    np.savez_compressed(os.path.join(HERE, "pivot.npz"), **out)


def pivot65_section():
    """---- 15b. nugget="pivot", n = 65 with TWO repeated design points (rank 63: dpstrf stops inside its first 64-column block) ----
    The inputs are case 1026 of `tests/tools/fuzz_parity.py 1500 311` (UniformSqExp, D = 2), the one class the randomised
    device-vs-oracle test reports (missing 5).  The two skipped rows keep what LAPACK left below their diagonal
    (linalg/cholesky.py:315-325 only replaces the diagonal): rounding residue ~1e-15 .. 1e-21, divided by replacement diagonals of
    7e-6 and 1e-7 in the forward substitution -- so y[63], y[64] and with them the log-posterior depend on the LAPACK build
    (this file: the reference under its conda environment's MKL).  Stored next to the raw outputs: the parts that do NOT depend on
    it (pivot order, the leading 63 x 63 block, log-determinant, the quadratic form of the 63 accepted pivots)."""
    from mogp_emulator.Kernel import UniformSqExp
    src = np.load(os.path.join(HERE, "pivot65_inputs.npz"))
    X, t, Xs, theta = src["X"], src["t"], src["Xs"], src["theta"]
    gp = GaussianProcess(X, t, kernel=UniformSqExp(), nugget="pivot", priors=GPPriors(n_corr=1, nugget_type="pivot"))
    gp.fit(theta)
    L, P = gp.Kinv.L, np.asarray(gp.Kinv.P, dtype=np.int64)
    y = np.linalg.solve(np.tril(L), t[P])
    mean, var, _ = gp.predict(Xs)
    out = dict(X=X, t=t, Xs=Xs, theta=theta, L=L, P=P, logpost=np.array(gp.current_logpost), Kinv_t=gp.Kinv_t, y=y,
               logdet=np.array(2. * np.sum(np.log(np.diag(L)))), quad_lead=np.array(y[:63] @ y[:63]), quad=np.array(y @ y),
               mean=mean, var=var, grad=gp.logpost_deriv(theta))
    np.savez_compressed(os.path.join(HERE, "pivot65.npz"), **out)


def validation_section():
    """---- 16. validation.py (consumer of predict(full_cov=True)): standard / pivoted errors, Mahalanobis ----"""
    from mogp_emulator.validation import mahalanobis, standard_errors, pivoted_errors, generate_mahal_dist
    X, T, Xv = synth(16, 60, 2, 3, 25)
    rng = np.random.default_rng(1616)
    w = rng.normal(size=(3, 2))
    Tv = np.stack([np.sin(2 * np.pi * Xv @ w_k / np.sqrt(2)) + 0.1 * (Xv ** 2) @ np.abs(w_k) for w_k in w]) + 0.02 * rng.normal(size=(3, 25))
    out = dict(X=X, T=T, Xv=Xv, Tv=Tv)
    for kern in KERNELS:
        for mode, nugget in (("fixed", 1.e-4), ("fit", "fit")):
            theta = [2.5, 3.0, 0.1] + ([np.log(3.e-4)] if mode == "fit" else [])
            nt = nugget if isinstance(nugget, str) else "fixed"
            pre = "%s_%s_" % (kern, mode)
            out[pre + "theta"] = np.array(theta)
            gp = GaussianProcess(X, T[0], kernel=KERNELS[kern](), nugget=nugget, priors=weak(2, nt))
            gp.fit(np.array(theta))
            e, P = standard_errors(gp, Xv, Tv[0])
            out[pre + "std_err"], out[pre + "std_P"] = e, np.asarray(P, dtype=np.int64)
            e, P = pivoted_errors(gp, Xv, Tv[0])
            out[pre + "piv_err"], out[pre + "piv_P"] = e, np.asarray(P, dtype=np.int64)
            out[pre + "mahal"] = np.array(mahalanobis(gp, Xv, Tv[0]))
            out[pre + "mahal_scaled"] = np.array(mahalanobis(gp, Xv, Tv[0], scaled=True))
            d = generate_mahal_dist(gp, Xv)
            out[pre + "dist_args"] = np.array([d.kwds["dfn"], d.kwds["dfd"], d.kwds["scale"]], dtype=float)
            mo = MultiOutputGP(X, T, kernel=KERNELS[kern](), nugget=nugget, priors=weak(2, nt))
            for em in mo.emulators:
                em.fit(np.array(theta))
            se = standard_errors(mo, Xv, Tv)
            pe = pivoted_errors(mo, Xv, Tv)
            out[pre + "mo_std_err"] = np.stack([x[0] for x in se]); out[pre + "mo_std_P"] = np.stack([np.asarray(x[1], dtype=np.int64) for x in se])
            out[pre + "mo_piv_err"] = np.stack([x[0] for x in pe]); out[pre + "mo_piv_P"] = np.stack([np.asarray(x[1], dtype=np.int64) for x in pe])
            out[pre + "mo_mahal"] = np.array(mahalanobis(mo, Xv, Tv))
            out[pre + "mo_mahal_scaled"] = np.array(mahalanobis(mo, Xv, Tv, scaled=True))
    np.savez_compressed(os.path.join(HERE, "validation.npz"), **out)


def tsunami_section():
    """---- 17. the reference's tsunami benchmark data (benchmarks/tsunamidata.npz, copied as a fixture): MAP fits of the
    first four outputs with the reference's default settings (default priors, adaptive nugget, 15 starts) ----"""
    f = np.load(os.path.join(HERE, "tsunamidata.npz"))
    X, T = f["inputs"], f["targets"][:4]
    np.random.seed(1717)
    mo = MultiOutputGP(X, T)
    mo = fit_GP_MAP(mo)
    rng = np.random.default_rng(17)
    Xs = X[rng.choice(X.shape[0], 20, replace=False)] + 0.01 * rng.normal(size=(20, X.shape[1]))
    mean, var, _ = mo.predict(Xs)
    np.savez_compressed(os.path.join(HERE, "tsunami_fit.npz"), theta=np.stack([em.theta.get_data() for em in mo.emulators]),
                        logpost=np.array([em.current_logpost for em in mo.emulators]), nugget=np.array([em.nugget for em in mo.emulators]),
                        Xs=Xs, mean=mean, var=var)


def branin_pivot_section():
    """---- 18. the reference's pivot benchmark (benchmarks/benchmark_pivot.py): 2-D Branin function on Latin-hypercube
    designs with one duplicated point, MAP fits with nugget="adaptive" and nugget="pivot", accuracy on random test points ----"""
    from mogp_emulator import LatinHypercubeDesign, MonteCarloDesign
    from scipy.stats import uniform

    def branin(x):
        x1, x2 = x[:, 0], x[:, 1]
        a, b, c, r, s, t = 1., 5.1 / 4. / np.pi ** 2, 5. / np.pi, 6., 10., 1. / 8. / np.pi
        return a * (x2 - b * x1 ** 2 + c * x1 - r) ** 2 + s * (1. - t) * np.cos(x1) + s

    space = [uniform(loc=-5., scale=15.).ppf, uniform(loc=0., scale=15.).ppf]
    np.random.seed(1818)
    out = {}
    testing = MonteCarloDesign(space).sample(100)
    out["testing"], out["test_targets"] = testing, branin(testing)
    for n_sim in (5, 10, 15, 20, 25, 30):
        X = LatinHypercubeDesign(space).sample(n_sim)
        X = np.vstack([X, X[:1]])
        t = branin(X)
        pre = "n%d_" % n_sim
        out[pre + "X"], out[pre + "t"] = X, t
        for tag in ("adaptive", "pivot"):
            gp = fit_GP_MAP(GaussianProcess(X, t, nugget=tag))
            if gp.theta.get_data() is None:
                continue
            mean, var, _ = gp.predict(testing, deriv=False, unc=True)
            out[pre + tag + "_theta"] = gp.theta.get_data()
            out[pre + tag + "_logpost"] = np.array(gp.current_logpost)
            out[pre + tag + "_mean"], out[pre + tag + "_var"] = mean, var
            if tag == "pivot":
                out[pre + "pivot_P"] = np.asarray(gp.Kinv.P, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "branin_pivot.npz"), **out)


def main():
    if sys.argv[1:] == ["branin"]:
        branin_pivot_section()
        return
    if sys.argv[1:] == ["tsunami"]:
        tsunami_section()
        return
    if sys.argv[1:] == ["pivot"]:
        pivot_section()
        return
    if sys.argv[1:] == ["pivot65"]:
        pivot65_section()
        return
    if sys.argv[1:] == ["validation"]:
        validation_section()
        return
    # ---- 1. the 2x3 fixture of tests/test_GaussianProcess.py:16-22, 556 ------------------------
    X = np.array([[1., 2., 3.], [4., 5., 6.]])
    t = np.array([2., 4.])
    Xs = np.array([[2., 3., 4.]])
    d = dict(X=X, t=t, Xs=Xs)
    for kern in KERNELS:
        for name, theta in (("ones", np.ones(4)), ("zeros", np.zeros(4))):
            _, o = run_gp(X, t, kern, 0., theta, Xs)
            for k, v in o.items():
                d["%s_%s_%s" % (kern, name, k)] = v
    np.savez(os.path.join(HERE, "fixture_2x3.npz"), **d)

    # ---- 2. 11x11 grid of tests/test_GaussianProcess.py:632-650 ---------------------------------
    xg, yg = np.meshgrid(np.linspace(0., 4., 11), np.linspace(0., 4., 11))
    X = np.stack([xg.ravel(), yg.ravel()], axis=1)
    t = np.exp(-0.5 * ((X[:, 0] - 2.) ** 2 + (X[:, 1] - 3.) ** 2))
    rng = np.random.default_rng(7)
    Xs = rng.uniform(0., 4., (32, 2))
    d = dict(X=X, t=t, Xs=Xs)
    for kern in KERNELS:
        for mode, nugget, theta in (("fixed", 1.e-6, [-1., -1., -2.]),
                                    ("fit", "fit", [-1., -1., -2., np.log(1.e-6)]),
                                    ("adaptive", "adaptive", [-1., -1., -2.])):
            _, o = run_gp(X, t, kern, nugget, theta, Xs)
            d["%s_%s_theta" % (kern, mode)] = np.array(theta)
            for k, v in o.items():
                if k == "K":
                    continue
                d["%s_%s_%s" % (kern, mode, k)] = v
    np.savez_compressed(os.path.join(HERE, "grid11.npz"), **d)

    # ---- 3. kernel known answers / derivative tensors --------------------------------------------
    rng = np.random.default_rng(3)
    x1 = rng.normal(size=(7, 3))
    x2 = rng.normal(size=(5, 3))
    th = np.array([0.3, -0.7, 1.1])  # reference kernels take D correlation params (Kernel.py:63-74)
    d = dict(x1=x1, x2=x2, theta=th)
    for kern, cls in KERNELS.items():
        k = cls()
        d[kern + "_r2"] = k.calc_r2(x1, x2, th)
        d[kern + "_K"] = k.kernel_f(x1, x2, th)
        d[kern + "_dKdtheta"] = k.kernel_deriv(x1, x2, th)
    # closed forms from tests/test_Kernel.py:721-754, 1021-1061
    d["closed_x"] = np.array([[1.], [2.]])
    d["closed_y"] = np.array([[2.], [3.]])
    d["closed_theta"] = np.zeros(1)
    for kern, cls in KERNELS.items():
        d[kern + "_closed_K"] = cls().kernel_f(d["closed_x"], d["closed_y"], d["closed_theta"])
    np.savez(os.path.join(HERE, "kernels.npz"), **d)

    # ---- 4. Cholesky known answers, tests/test_linalg.py:103-154 ---------------------------------
    wiki = np.array([[4., 12., -16.], [12., 37., -43.], [-16., -43., 98.]])
    Lw, jw = jit_cholesky(wiki)
    sing = np.array([[1., 1.], [1., 1.]])
    Ls, js = jit_cholesky(sing)
    np.savez(os.path.join(HERE, "cholesky.npz"), wiki=wiki, wiki_L=Lw, wiki_jitter=jw,
             sing=sing, sing_L=Ls, sing_jitter=js, wiki_fixed_L=fixed_cholesky(wiki))

    # ---- 5. priors: log densities, gradients, defaults ------------------------------------------
    xs = np.array([0.05, 0.3, 1.0, 3.0, 12.5])
    d = dict(x=xs)
    for nm, cls in (("invgamma", InvGammaPrior), ("gamma", GammaPrior), ("lognormal", LogNormalPrior)):
        p = cls(2., 2.)
        d[nm + "_2_2_logp"] = np.array([p.logp(x) for x in xs])
        d[nm + "_2_2_dlogpdx"] = np.array([p.dlogpdx(x) for x in xs])
        p = cls(0.84, 0.0017)
        d[nm + "_b_logp"] = np.array([p.logp(x) for x in xs])
    # default priors, Priors.py:85-152 (pinning I/O: SURVEY.md section 8b)
    Xd = np.random.default_rng(0).uniform(0, 1, (2000, 10))
    dp = GPPriors.default_priors(Xd, 10, "fit")
    d["default_X_seed"] = np.array(0)
    d["default_corr_shape"] = np.array([p.shape for p in dp.corr])
    d["default_corr_scale"] = np.array([p.scale for p in dp.corr])
    d["default_nugget"] = np.array([float(dp.nugget.shape), float(dp.nugget.scale)])
    d["default_min_spacing"] = np.array([min_spacing(c) for c in Xd.T])
    d["default_max_spacing"] = np.array([max_spacing(c) for c in Xd.T])
    # GPPriors.logp / dlogpdtheta on a 3-corr "fit" parameter vector
    pri = GPPriors(corr=[InvGammaPrior(2., 1.), GammaPrior(3., 0.5), LogNormalPrior(0.7, 1.3)],
                   cov=GammaPrior(2., 3.), nugget=InvGammaPrior(3.3, 4.3e-7), nugget_type="fit")
    gpp = GPParams(n_mean=0, n_corr=3, nugget="fit")
    thp = np.array([0.4, -1.2, 2.0, 0.7, np.log(2.e-7)])
    gpp.set_data(thp)
    d["gppriors_theta"] = thp
    d["gppriors_logp"] = np.array(pri.logp(gpp))
    d["gppriors_dlogp"] = pri.dlogpdtheta(gpp)
    np.savez(os.path.join(HERE, "priors.npz"), **d)

    # ---- 6. variance-stability regression, tests/test_GaussianProcess.py:1144-1161 --------------
    x = np.linspace(0., 5., 21)
    y = x ** 2
    xt = np.linspace(0., 5., 101)
    gp = GaussianProcess(x, y, nugget=1.e-8, priors=weak(1, "fixed"))  # predictions are prior-independent
    th = np.array([-7.352408190715323, 15.041447753599755])
    gp.fit(th)
    mean, var, _ = gp.predict(xt)
    np.savez(os.path.join(HERE, "var_stability.npz"), x=x, y=y, xt=xt, theta=th, mean=mean, var=var,
             logpost=gp.current_logpost)

    # ---- 7. medium synthetic configs (C1 and a d=10 case), SURVEY.md section 8c item 7 ----------
    for tag, cid, n, dd in (("c1_n200_d4", 1, 200, 4), ("n500_d10", 6, 500, 10)):
        X, T, Xs = synth(cid, n, dd, 2, 256)
        out = dict(X=X, T=T, Xs=Xs)
        corr = -2. * np.log(0.3 * np.sqrt(dd))
        for kern in KERNELS:
            for mode, nugget in (("fixed", 1.e-6), ("fit", "fit"), ("adaptive", "adaptive")):
                theta = [corr] * dd + [0.]
                if mode == "fit":
                    theta = theta + [np.log(1.e-4)]
                _, o = run_gp(X, T[0], kern, nugget, theta, Xs)
                pre = "%s_%s_" % (kern, mode)
                out[pre + "theta"] = np.array(theta)
                K = o.pop("K")
                L = o.pop("L")
                out[pre + "K_sum"] = np.array(K.sum())
                out[pre + "K_rows"] = K[::37, ::41]
                out[pre + "L_diag"] = np.diag(L)
                out[pre + "L_rows"] = L[::37, ::41]
                for k, v in o.items():
                    out[pre + k] = v
        # default (non-weak) priors: logpost + grad with the prior terms
        gp = GaussianProcess(X, T[1], nugget="fit")
        theta = np.array([corr] * dd + [0.1, np.log(3.e-7)])
        gp.fit(theta)
        out["defprior_theta"] = theta
        out["defprior_logpost"] = np.array(gp.current_logpost)
        out["defprior_grad"] = gp.logpost_deriv(theta)
        out["defprior_corr_shape"] = np.array([p.shape for p in gp.priors.corr])
        out["defprior_corr_scale"] = np.array([p.scale for p in gp.priors.corr])
        out["defprior_nugget"] = np.array([float(gp.priors.nugget.shape), float(gp.priors.nugget.scale)])
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)

    # ---- 8. multi-output: 4 emulators sharing X --------------------------------------------------
    X, T, Xs = synth(8, 60, 3, 4, 40)
    thetas = np.array([[0.5, 0.2, -0.3, 0.1], [1.0, 1.0, 1.0, -0.5], [-0.2, 0.4, 0.9, 0.3], [0., 0., 0., 0.]])
    out = dict(X=X, T=T, Xs=Xs, thetas=thetas)
    means, vs, lps, grads = [], [], [], []
    for k in range(4):
        _, o = run_gp(X, T[k], "SquaredExponential", 1.e-6, thetas[k], Xs)
        means.append(o["mean"]); vs.append(o["var"]); lps.append(o["logpost"]); grads.append(o["grad"])
    out.update(mean=np.array(means), var=np.array(vs), logpost=np.array(lps), grad=np.array(grads))
    np.savez_compressed(os.path.join(HERE, "mogp4.npz"), **out)

    # ---- 9. fit_GP_MAP end point on C1 (objective value only; trajectory parity is unpinned) ----
    X, T, Xs = synth(1, 200, 4, 2, 256)
    np.random.seed(1234)
    gp = GaussianProcess(X, T[0], nugget=1.e-6)
    theta0 = np.array([-2. * np.log(0.3 * 2.)] * 4 + [0.])
    gp = fit_GP_MAP(gp, n_tries=1, theta0=theta0)
    mean, var, _ = gp.predict(Xs)
    np.savez_compressed(os.path.join(HERE, "fitmap_c1.npz"), X=X, t=T[0], Xs=Xs, theta0=theta0,
                        theta_hat=gp.theta.get_data(), logpost_hat=np.array(gp.current_logpost),
                        mean=mean, var=var,
                        corr_shape=np.array([p.shape for p in gp.priors.corr]),
                        corr_scale=np.array([p.scale for p in gp.priors.corr]))
    # ---- 10. analytic mean function (CPU semantics, weak mean priors): SURVEY 8f row 1 -----------
    X, T, Xs = synth(10, 150, 3, 1, 50)
    tlin = T[0] + 2.0 + 1.5 * X[:, 0] - 0.7 * X[:, 2] ** 2
    out = dict(X=X, t=tlin, Xs=Xs)
    for tag, formula in (("lin", "x[0]"), ("two", "x[0]+x[2]"), ("const", "1"), ("quad", "x[0]+I(x[2]**2)")):
        for kern in KERNELS:
            for mode, nugget in (("fixed", 1.e-5), ("fit", "fit"), ("adaptive", "adaptive")):
                theta = [0.4, -0.2, 0.7, 0.3] + ([np.log(2.e-4)] if mode == "fit" else [])
                nt = nugget if isinstance(nugget, str) else "fixed"
                gp = GaussianProcess(X, tlin, mean=formula, kernel=KERNELS[kern](), nugget=nugget, priors=weak(3, nt))
                gp.fit(np.array(theta))
                pre = "%s_%s_%s_" % (tag, kern, mode)
                out[pre + "theta"] = np.array(theta)
                out[pre + "logpost"] = np.array(gp.current_logpost)
                out[pre + "grad"] = gp.logpost_deriv(np.array(theta))
                out[pre + "beta"] = np.array(gp.theta.mean)
                out[pre + "nugget"] = np.array(gp.nugget)
                out[pre + "Kinv_t_mean"] = gp.Kinv_t_mean
                mean, var, _ = gp.predict(Xs)
                out[pre + "mean"] = mean
                out[pre + "var"] = var
                out[pre + "dm"] = gp.get_design_matrix(X)[:5]
    np.savez_compressed(os.path.join(HERE, "meanfunc.npz"), **out)

    # ---- 11. predict(full_cov=True): SURVEY 8f row 3 ------------------------------------------------
    X, T, Xs = synth(11, 180, 3, 1, 70)
    tq = T[0] + 1.0 - 0.8 * X[:, 1]
    out = dict(X=X, t=tq, Xs=Xs)
    for tag, formula in (("zero", None), ("lin", "x[1]")):
        for kern in KERNELS:
            for mode, nugget in (("fixed", 1.e-5), ("fit", "fit")):
                theta = [0.3, 0.1, -0.4, 0.2] + ([np.log(3.e-4)] if mode == "fit" else [])
                nt = nugget if isinstance(nugget, str) else "fixed"
                gp = GaussianProcess(X, tq, mean=formula, kernel=KERNELS[kern](), nugget=nugget, priors=weak(3, nt))
                gp.fit(np.array(theta))
                pre = "%s_%s_%s_" % (tag, kern, mode)
                out[pre + "theta"] = np.array(theta)
                mean, cov, _ = gp.predict(Xs, full_cov=True)
                out[pre + "mean"] = mean
                out[pre + "cov"] = cov
                out[pre + "cov_nonug"] = gp.predict(Xs, full_cov=True, include_nugget=False)[1]
                out[pre + "var"] = gp.predict(Xs)[1]
    np.savez_compressed(os.path.join(HERE, "fullcov.npz"), **out)

    # ---- 12. consumers of predict: HistoryMatching implausibility, MICE criterion ----
    from mogp_emulator.HistoryMatching import HistoryMatching
    from mogp_emulator.SequentialDesign import MICEFastGP
    from mogp_emulator import MultiOutputGP
    X, T, Xs = synth(12, 90, 2, 3, 150)
    thetas = np.array([[1.2, 0.4, 0.1], [0.3, 1.5, -0.3], [2.0, 2.2, 0.5]])
    mo = MultiOutputGP(X, T, nugget=1.e-4, priors=weak(2, "fixed"))
    for k in range(3):
        mo.emulators[k].fit(thetas[k])
    obs = [np.array([0.3, -0.2, 0.6]), np.array([0.01, 0.02, 0.005])]
    out = dict(X=X, T=T, Xs=Xs, thetas=thetas, obs=obs[0], obs_var=obs[1], disc=np.array([0.1, 0., 0.3]))
    for rank in range(3):
        hm = HistoryMatching(gp=mo, obs=obs, coords=Xs)
        out["I_rank%d" % rank] = hm.get_implausibility(rank=rank)
        hm = HistoryMatching(gp=mo, obs=obs, coords=Xs)
        out["I_disc_rank%d" % rank] = hm.get_implausibility(out["disc"], rank=rank)
    hm = HistoryMatching(gp=mo, obs=obs, coords=Xs, threshold=2.5)
    out["NROY"] = np.array(hm.get_NROY(0.05, rank=1))
    out["RO"] = np.array(hm.get_RO(0.05, rank=1))
    hm1 = HistoryMatching(gp=mo.emulators[1], obs=[-0.2, 0.02], coords=Xs)
    out["I_single"] = hm1.get_implausibility(0.07)
    # MICE: base GP on X, candidates = first 60 query points.  MICEFastGP.fast_predict reads ``self.L``, an
    # attribute the refactored GaussianProcess no longer has (it lives in ``self.Kinv.L``; the reference's own
    # MICE tests are skipped for that reason, tests/test_SequentialDesign.py:866-945): the attribute is supplied
    # here so that the reference's arithmetic itself produces the vectors.
    class MICEFastGP(MICEFastGP):
        L = property(lambda self: self.Kinv.L)
    known = MICEFastGP(np.reshape([1., 2., 3., 4], (4, 1)), [1., 1., 1., 1.])
    known.theta = np.array([0., -1.])
    out["mice_known_answer"] = np.array(known.fast_predict(3))     # 1.191061906777163 in tests/test_SequentialDesign.py:935
    cand = Xs[:60]
    base = mo.emulators[0]
    for nugget_s in (1., 10.):
        fast = MICEFastGP(cand, np.ones(len(cand)), nugget=base.theta.nugget * nugget_s)
        # MICEDesign._eval_metric assigns the base GP's GPParams OBJECT (SequentialDesign.py:946-947), which
        # replaces the candidate GP's own nugget by the base nugget: nugget_s is then silently ignored.  That
        # variant is recorded as "*_aliased"; the documented behaviour (nugget = base nugget * nugget_s) is
        # obtained by assigning the parameter VALUES instead.
        fast.theta = base.theta
        out["mice_unc2_s%d_aliased" % int(nugget_s)] = np.array([fast.fast_predict(c)[0] for c in range(len(cand))])
        fast = MICEFastGP(cand, np.ones(len(cand)), nugget=base.theta.nugget * nugget_s)
        fast.theta = base.theta.get_data()
        unc2 = np.array([fast.fast_predict(c)[0] for c in range(len(cand))])
        unc1 = base.predict(cand, unc=True, deriv=False)[1]
        out["mice_unc1"] = unc1
        out["mice_unc2_s%d" % int(nugget_s)] = unc2
        out["mice_crit_s%d" % int(nugget_s)] = unc1 / unc2
    np.savez_compressed(os.path.join(HERE, "consumers.npz"), **out)

    # ---- 13. CPU-only kernels: UniformSqExp, UniformMat52, ProductMat52 -------------
    from mogp_emulator.Kernel import UniformSqExp, UniformMat52, ProductMat52
    X, T, Xs = synth(13, 160, 3, 1, 60)
    out = dict(X=X, t=T[0], Xs=Xs)
    for name, cls, nc in (("UniformSqExp", UniformSqExp, 1), ("UniformMat52", UniformMat52, 1), ("ProductMat52", ProductMat52, 3)):
        k = cls()
        corr = np.array([0.7, -0.3, 1.1])[:nc]
        out[name + "_kf"] = k.kernel_f(X[:7], Xs[:5], corr)
        out[name + "_kd"] = k.kernel_deriv(X[:7], Xs[:5], corr)
        for mode, nugget in (("fixed", 1.e-5), ("fit", "fit")):
            theta = list(corr) + [0.25] + ([np.log(2.e-4)] if mode == "fit" else [])
            nt = nugget if isinstance(nugget, str) else "fixed"
            gp = GaussianProcess(X, T[0], kernel=k, nugget=nugget, priors=weak(nc, nt))
            gp.fit(np.array(theta))
            pre = "%s_%s_" % (name, mode)
            out[pre + "theta"] = np.array(theta)
            out[pre + "logpost"] = np.array(gp.current_logpost)
            out[pre + "grad"] = gp.logpost_deriv(np.array(theta))
            out[pre + "Kinv_t"] = gp.Kinv_t
            mean, var, _ = gp.predict(Xs)
            out[pre + "mean"] = mean
            out[pre + "var"] = var
        # default priors (one InvGamma per correlation parameter; the uniform kernels pool all inputs)
        gp = GaussianProcess(X, T[0], kernel=k, nugget="fit")
        theta = np.array(list(corr) + [0.25, np.log(2.e-4)])
        gp.fit(theta)
        out[name + "_defprior_logpost"] = np.array(gp.current_logpost)
        out[name + "_defprior_grad"] = gp.logpost_deriv(theta)
    np.savez_compressed(os.path.join(HERE, "kernels_cpuonly.npz"), **out)

    # ---- 14. analytic mean with informative mean priors (MeanPriors mean / cov), Priors.py:423-581 ------
    from mogp_emulator.Priors import MeanPriors
    X, T, Xs = synth(10, 150, 3, 1, 50)
    tlin = T[0] + 2.0 + 1.5 * X[:, 0] - 0.7 * X[:, 2] ** 2
    out = dict(X=X, t=tlin, Xs=Xs)
    Bfull = np.array([[2.0, 0.3, -0.1], [0.3, 1.0, 0.2], [-0.1, 0.2, 0.5]])
    cases = {"scalar": ("x[0]", [1.5, 1.0], 4.0), "vector": ("x[0]+I(x[2]**2)", [2.0, 1.0, -1.0], [3.0, 1.0, 0.25]),
             "matrix": ("x[0]+I(x[2]**2)", [2.0, 1.0, -1.0], Bfull), "tight": ("1", [0.5], 1.e-3)}
    for tag, (formula, b, cov) in cases.items():
        for kern in KERNELS:
            for mode, nugget in (("fixed", 1.e-5), ("fit", "fit")):
                theta = [0.4, -0.2, 0.7, 0.3] + ([np.log(2.e-4)] if mode == "fit" else [])
                nt = nugget if isinstance(nugget, str) else "fixed"
                pri = GPPriors(mean=MeanPriors(mean=b, cov=cov), n_corr=3, nugget_type=nt)
                gp = GaussianProcess(X, tlin, mean=formula, kernel=KERNELS[kern](), nugget=nugget, priors=pri)
                gp.fit(np.array(theta))
                pre = "%s_%s_%s_" % (tag, kern, mode)
                out[pre + "theta"] = np.array(theta)
                out[pre + "b"] = np.array(b, dtype=float)
                out[pre + "cov"] = np.array(cov, dtype=float)
                out[pre + "logpost"] = np.array(gp.current_logpost)
                out[pre + "grad"] = gp.logpost_deriv(np.array(theta))
                out[pre + "beta"] = np.array(gp.theta.mean)
                out[pre + "Kinv_t_mean"] = gp.Kinv_t_mean
                mean, var, _ = gp.predict(Xs)
                out[pre + "mean"] = mean
                out[pre + "var"] = var
                out[pre + "cov_full"] = gp.predict(Xs, full_cov=True)[1]
    np.savez_compressed(os.path.join(HERE, "meanpriors.npz"), **out)
    pivot_section()
    pivot65_section()
    validation_section()
    tsunami_section()
    branin_pivot_section()
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
