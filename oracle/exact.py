"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

The likelihood part of the negative log-posterior (GaussianProcess.py:657-685 with H = n x 0:
0.5 (t^T K^-1 t + log|K| + n log 2 pi)) of a GIVEN fp64 matrix K and targets t, evaluated in 80-bit long double
(numpy longdouble on x86-64: 64-bit mantissa, ~2000 x finer than fp64) by an unblocked-in-the-block Cholesky without BLAS.

Why it exists: for the ill-conditioned matrices GP fits produce (cond(K) ~ 1e9 with a 1e-6 nugget) two backward-stable fp64
factorisations of the same K differ from the exact value -- and from each other -- by ~1e-3 cond(K) eps.  LAPACK's own result sits
1e-11 ... 5e-10 (relative) from the exact one at n = 1500, d = 6 (tools/logpost_truth.py), so "rtol 1e-10 against LAPACK" is a
tighter bar than LAPACK itself meets.  The tests that run such matrices compare the device result AND the fp64 oracle with this
value, to a tolerance stated in units of cond(K) eps.
"""
import numpy as np


def loglike_longdouble(K, T, nb=64):
    """K: (n, n) fp64 SPD matrix (nugget already on the diagonal); T: (B, n) targets.  Returns a longdouble array (B,)."""
    n = K.shape[0]
    A = np.asarray(K, dtype=np.longdouble)
    L = np.zeros_like(A)
    for c in range(0, n, nb):
        e = min(c + nb, n)
        S = A[c:, c:e] - L[c:, :c] @ L[c:e, :c].T
        for j in range(e - c):
            S[j, j] = np.sqrt(S[j, j])
            S[j + 1:, j] /= S[j, j]
            S[j + 1:, j + 1:e - c] -= np.outer(S[j + 1:, j], S[j + 1:e - c, j])
        L[c:, c:e] = S
        L[c:e, c:e] = np.tril(S[:e - c])
    logdet = 2 * np.sum(np.log(np.diag(L)))
    T = np.atleast_2d(np.asarray(T, dtype=np.longdouble))
    Y = T.T.copy()                                   # forward substitution for all right-hand sides at once
    for i in range(n):
        Y[i] = (Y[i] - L[i, :i] @ Y[:i]) / L[i, i]
    return 0.5 * (np.sum(Y * Y, axis=0) + logdet + n * np.log(np.longdouble(2) * np.pi))


def min_pivot_longdouble(K):
    """Smallest pivot d_j = K_jj - sum_k L_jk^2 of the Cholesky factorisation of the fp64 matrix K carried out in 80-bit long double
    (the pivots of K itself up to ~1e-19 relative: what ANY fp64 factorisation perturbs by its rounding).  Stops at the first
    non-positive pivot and returns it: K is then not positive definite as a matrix of real numbers.  Used to classify the points of
    the adaptive-nugget sweep (linalg/cholesky.py:234-281: plain dpotrf first, jitter only when it fails) into clearly definite,
    clearly indefinite and the knife-edge band in between, where fp64 factorisations with different summation orders may decide
    differently (tests/test_gpu_parity.py::test_adaptive_nugget_decision_sweep_across_the_knife_edge)."""
    n = K.shape[0]
    A = np.array(K, dtype=np.longdouble)
    dmin = np.longdouble(np.inf)
    for j in range(n):
        d = A[j, j]
        dmin = min(dmin, d)
        if not d > 0:
            return float(dmin)
        A[j + 1:, j] /= np.sqrt(d)
        A[j + 1:, j + 1:] -= np.outer(A[j + 1:, j], A[j + 1:, j])
    return float(dmin)


def knife_edge_class(K, c=8.0):
    """Classify an fp64 matrix for the adaptive-nugget decision (linalg/cholesky.py:234-281: jitter only when the plain factorisation fails) by
    the pivots of its Cholesky factorisation in 80-bit long double: tau = c * max(n, 32) * eps * max K_ii is what the rounding of ANY fp64
    factorisation (LAPACK's blocked dpotrf, the device's 4-column groups with explicit 4 x 4 inverses) may move a pivot by, and once a pivot
    is below tau the following ones are amplified rounding noise in every implementation.  So the FIRST exact pivot below tau decides:
      none                    -> ("definite", smallest pivot / tau): every correct implementation factors without jitter;
      first one <= -tau       -> ("indefinite", that pivot / tau):   every correct implementation must jitter;
      first one in (-tau, tau)-> ("band", that pivot / tau):         either decision is a correct execution of the reference's algorithm."""
    n = K.shape[0]
    A = np.array(K, dtype=np.longdouble)
    tau = np.longdouble(c * max(n, 32) * 2.0 ** -52) * np.longdouble(np.max(np.diag(K)))
    dmin = np.longdouble(np.inf)
    for j in range(n):
        d = A[j, j]
        if d < tau:
            return ("indefinite" if d <= -tau else "band"), float(d / tau)
        dmin = min(dmin, d)
        A[j + 1:, j] /= np.sqrt(d)
        A[j + 1:, j + 1:] -= np.outer(A[j + 1:, j], A[j + 1:, j])
    return "definite", float(dmin / tau)


def cond_eps(K):
    """cond_2(K) * 2^-52 of a symmetric positive definite fp64 matrix."""
    w = np.linalg.eigvalsh(K)
    return float(w[-1] / w[0]) * 2.0 ** -52


def loglike_from_factor_longdouble(L, t):
    """The same likelihood term from a GIVEN fp64 Cholesky factor L (lower, n x n) and targets t (n): forward substitution and
    log-determinant in 80-bit long double, O(n^2).  What it isolates: the value the factor itself stands for -- exact for the matrix
    L L^T = K + E -- so that (device value - this) is the error of the device's solve / reduction arithmetic alone, and the difference
    of this quantity between two factors of the same K is the sensitivity of the likelihood to their backward errors E (~ cond eps)."""
    n = L.shape[0]
    y = np.zeros(n, dtype=np.longdouble)
    tl = np.asarray(t, dtype=np.longdouble)
    for i in range(n):
        row = np.asarray(L[i, :i], dtype=np.longdouble)
        y[i] = (tl[i] - row @ y[:i]) / np.longdouble(L[i, i])
    logdet = 2 * np.sum(np.log(np.asarray(np.diag(L), dtype=np.longdouble)))
    return 0.5 * (np.sum(y * y) + logdet + n * np.log(np.longdouble(2) * np.pi))


def factor_backward_error(K_rows, L, probes=4, seed=0):
    """max over a few random vectors v of ||(K - L L^T) v||_inf / (||K||_inf ||v||_inf), the products in long double; K_rows(i0, i1) returns
    rows [i0, i1) of the fp64 matrix K (so that K need not be held as one array)."""
    n = L.shape[0]
    rng = np.random.default_rng(seed)
    V = rng.standard_normal((n, probes))
    Vl = np.asarray(V, dtype=np.longdouble)
    W = np.zeros_like(Vl)                                # L^T V
    for i0 in range(0, n, 512):
        i1 = min(i0 + 512, n)
        W[:i1] += np.asarray(L[i0:i1, :i1], dtype=np.longdouble).T @ Vl[i0:i1]
    err, knorm = 0., 0.
    for i0 in range(0, n, 512):
        i1 = min(i0 + 512, n)
        Kr = K_rows(i0, i1)
        r = np.asarray(Kr, dtype=np.longdouble) @ Vl - np.asarray(L[i0:i1, :i1], dtype=np.longdouble) @ W[:i1]
        err = max(err, float(np.max(np.abs(r))))
        knorm = max(knorm, float(np.max(np.sum(np.abs(Kr), axis=1))))
    return err / (knorm * float(np.max(np.abs(V))))
