"""Round 6: what would the Gram form of the squared distances, r2_ij = a_i + a_j - 2 sum_d (e_d x_id) x_jd (a K = d GEMM on the matrix cores),
cost in accuracy against the difference form sum_d e_d (x_id - x_jd)^2 that the reference (Kernel.py:444-485) and the device kernels use?  Host-only (NumPy, fp64;
the exact value of the likelihood of each K through oracle/exact.py in 80-bit long double is NOT used: the question is how far the two fp64 matrices and their
log-posteriors are apart, in units of the 1e-10 bar of the parity tests)."""
import os, sys
import numpy as np
import scipy.linalg as sl
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth


def kern(r2, kind):
    if kind == "SquaredExponential":
        return np.exp(-0.5 * r2)
    s = np.sqrt(5. * r2)
    return (1. + s + (5. / 3.) * r2) * np.exp(-s)


def logpost(K, t):
    L = sl.cholesky(K, lower=True)
    y = sl.solve_triangular(L, t, lower=True)
    return 0.5 * (y @ y + 2. * np.log(np.diag(L)).sum() + len(t) * np.log(2. * np.pi))


for tag, cid, n, d, kind, nug in (("C3-like", 2, 2000, 10, "SquaredExponential", 1e-6), ("C4-like", 4, 2000, 20, "Matern52", 1e-4), ("C5-like", 5, 2000, 8, "SquaredExponential", 1e-6)):
    X, T, _ = synth(cid, n, d, 1, 8)
    e = np.full(d, np.exp(-2. * np.log(0.3 * np.sqrt(d))))
    diff = X[:, None, :] - X[None, :, :]
    r2_ref = np.einsum("ijd,d->ij", diff * diff, e)
    res = {}
    for name, Xc in (("Gram form", X), ("Gram form, inputs centred", X - X.mean(axis=0))):
        a = (Xc * Xc) @ e
        r2 = a[:, None] + a[None, :] - 2. * (Xc * e) @ Xc.T
        np.fill_diagonal(r2, 0.)
        r2 = np.maximum(r2, 0.)
        res[name] = r2
    # the device's own variant of the difference form (inputs scaled by sqrt(e_d) first) for scale
    U = X * np.sqrt(e)
    du = U[:, None, :] - U[None, :, :]
    res["difference form on scaled inputs (device)"] = np.einsum("ijd,ijd->ij", du, du)
    K0 = kern(r2_ref, kind) + nug * np.eye(n)
    lp0 = logpost(K0, T[0])
    cond = np.linalg.cond(K0)
    print("%s: n=%d d=%d %s nugget %g, cond(K) %.1e, max a_i %.2f, log-posterior %.10g" % (tag, n, d, kind, nug, cond, ((X * X) @ e).max(), lp0))
    for name, r2 in res.items():
        K = kern(r2, kind) + nug * np.eye(n)
        off = ~np.eye(n, dtype=bool)
        print("   %-44s max |dr2| %.2e   max rel dK %.2e   rel d(log-posterior) %.2e" % (
            name, np.abs(r2 - r2_ref).max(), (np.abs(K - K0)[off] / K0[off]).max(), abs(logpost(K, T[0]) - lp0) / abs(lp0)))
