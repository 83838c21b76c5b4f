"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

NumPy/SciPy restatement of the mogp_emulator (v0.7.2) CPU ``GaussianProcess``
hot path: kernel-matrix build -> jittered Cholesky -> triangular solves ->
negative log-posterior (+ gradient) -> predictive mean / variance.  Every
function cites the reference file:line whose arithmetic it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker / the
reported CPU baseline.  The product (``mogp_emulator_amd``) never imports it.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz`` which were produced by importing the real reference in
the build container (``tests/golden/make_golden.py``), and against the
known-answer values held by the reference's own tests.

Scope: zero / fixed mean function (``mean=None``), SquaredExponential and
Matern52 kernels, nugget types adaptive / fit / fixed.  With ``mean=None`` the
reference's design matrix is n x 0, ``Ainv`` is 0 x 0 and every ``H``/``A`` term
in GaussianProcess.py:657-685, 745-778, 896-920 vanishes identically; that is
the branch restated here.
"""

import numpy as np
from scipy import linalg
from scipy.linalg import cho_factor, lapack, cho_solve
from scipy.special import gammaln

SQEXP = "SquaredExponential"
MAT52 = "Matern52"
# CPU-only kernels of the reference, Kernel.py:946-997
UNISQEXP = "UniformSqExp"
UNIMAT52 = "UniformMat52"
PRODMAT52 = "ProductMat52"
_BASE = {SQEXP: SQEXP, MAT52: MAT52, UNISQEXP: SQEXP, UNIMAT52: MAT52, PRODMAT52: MAT52}


def n_corr_of(kernel, D):
    """get_n_params: Kernel.py:16-32 (one per input), :229-242 (UniformKernel: 1)."""
    return 1 if kernel in (UNISQEXP, UNIMAT52) else D


# ----------------------------------------------------------------------------
# Kernel.py
# ----------------------------------------------------------------------------

def calc_r2(x1, x2, corr_raw):
    """Scaled squared distance, Kernel.py:444-485 (StationaryKernel.calc_r2).

    r2_ij = sum_d exp(theta_d) (x1_id - x2_jd)^2, built with the same
    (n1, n2, D) broadcast temporary and ``np.sum(axis=-1)`` as the reference.
    """
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    scale = np.exp(np.asarray(corr_raw, dtype=np.float64))
    diff = x1[:, np.newaxis, :] - x2[np.newaxis, :, :]
    r2 = np.sum(scale * diff ** 2, axis=-1)
    if np.any(np.isinf(r2)):
        raise FloatingPointError("Inf enountered in kernel distance computation")
    return r2


def calc_r2_chunked(x1, x2, corr_raw, rows=512):
    """Same arithmetic as :func:`calc_r2` but over row blocks of ``x1`` so the
    (n1, n2, D) temporary stays bounded (SURVEY.md section 8d memory guard).
    Per-entry operations and summation order are unchanged."""
    out = np.empty((x1.shape[0], x2.shape[0]))
    for s in range(0, x1.shape[0], rows):
        out[s:s + rows] = calc_r2(x1[s:s + rows], x2, corr_raw)
    return out


def calc_dr2dtheta(x1, x2, corr_raw):
    """d r2 / d theta_p = exp(theta_p) (x1_p - x2_p)^2, shape (D, n1, n2).
    Kernel.py:487-530."""
    scale = np.exp(np.asarray(corr_raw, dtype=np.float64))
    diff = x1[:, np.newaxis, :] - x2[np.newaxis, :, :]
    return np.transpose(scale * diff ** 2, (2, 0, 1))


def calc_K(r2, kernel=SQEXP):
    """k(r2).  SqExp: Kernel.py:772-791; Matern-5/2: Kernel.py:861-882."""
    r2 = np.asarray(r2)
    assert np.all(r2 >= 0.), "kernel distances must be positive"
    if kernel == SQEXP:
        return np.exp(-0.5 * r2)
    elif kernel == MAT52:
        return (1. + np.sqrt(5. * r2) + 5. / 3. * r2) * np.exp(-np.sqrt(5. * r2))
    raise ValueError("unknown kernel " + str(kernel))


def calc_dKdr2(r2, kernel=SQEXP):
    """dk/d(r2).  SqExp: Kernel.py:793-814; Matern-5/2: Kernel.py:884-906."""
    r2 = np.asarray(r2)
    assert np.all(r2 >= 0.), "kernel distances must be positive"
    if kernel == SQEXP:
        return -0.5 * np.exp(-0.5 * r2)
    elif kernel == MAT52:
        return -5. / 6. * (1. + np.sqrt(5. * r2)) * np.exp(-np.sqrt(5 * r2))
    raise ValueError("unknown kernel " + str(kernel))


def calc_r2_uniform(x1, x2, corr_raw):
    """UniformKernel.calc_r2, Kernel.py:296-336: one shared length scale exp(theta_0)."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    exp_theta = np.exp(np.asarray(corr_raw, dtype=np.float64))[0]
    r2 = np.sum(exp_theta * (x1[:, np.newaxis, :] - x2[np.newaxis, :, :]) ** 2, axis=-1)
    if np.any(np.isinf(r2)):
        raise FloatingPointError("Inf enountered in kernel distance computation")
    return r2


def calc_r2_product(x1, x2, corr_raw):
    """ProductKernel.calc_r2, Kernel.py:584-625: per-dimension scaled squared distances, shape (n1, n2, D)."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    exp_theta = np.exp(np.asarray(corr_raw, dtype=np.float64))
    r2 = exp_theta[np.newaxis, np.newaxis, :] * (x1[:, np.newaxis, :] - x2[np.newaxis, :, :]) ** 2
    if np.any(np.isinf(r2)):
        raise FloatingPointError("Inf enountered in kernel distance computation")
    return r2


def kernel_f(x1, x2, corr_raw, kernel=SQEXP):
    """Kernel.py:99-131 (kernel_f = calc_K(calc_r2)); ProductKernel.kernel_f :627-659."""
    if kernel == PRODMAT52:
        return np.prod(calc_K(calc_r2_product(x1, x2, corr_raw), MAT52), axis=-1)
    if kernel in (UNISQEXP, UNIMAT52):
        return calc_K(calc_r2_uniform(x1, x2, corr_raw), _BASE[kernel])
    return calc_K(calc_r2(x1, x2, corr_raw), kernel)


def kernel_deriv(x1, x2, corr_raw, kernel=SQEXP):
    """Kernel.py:133-173: dK/dtheta = dK/dr2 * dr2/dtheta, shape (n_corr, n1, n2).
    Uniform: dr2/dtheta_0 = r2 (:338-376).  Product (:661-705): the factor of dimension p is replaced by
    dK/dr2(r2_p) r2_p, all other factors are kept."""
    if kernel == PRODMAT52:
        r2 = calc_r2_product(x1, x2, corr_raw)
        D = r2.shape[-1]
        Kd = calc_K(r2, MAT52)
        diag = calc_dKdr2(r2, MAT52) * r2
        out = np.empty((D,) + r2.shape[:2])
        for p in range(D):
            f = Kd.copy()
            f[:, :, p] = diag[:, :, p]
            out[p] = np.prod(f, axis=-1)
        return out
    if kernel in (UNISQEXP, UNIMAT52):
        r2 = calc_r2_uniform(x1, x2, corr_raw)
        return (calc_dKdr2(r2, _BASE[kernel]) * r2)[np.newaxis]
    return calc_dKdr2(calc_r2(x1, x2, corr_raw), kernel) * calc_dr2dtheta(x1, x2, corr_raw)


def kernel_inputderiv(x1, x2, corr_raw, kernel=SQEXP):
    """d k(x1_i, x2_j) / d x1_id, shape (D, n1, n2).

    Not in the CPU reference (deriv deprecated, GaussianProcess.py:922-925);
    this is the quantity the reference GPU kernels ``*_cov_deriv_x_batch``
    (mogp_gpu/src/kernel.cu:69-100, 264-302) compute:
    dk/dr2 * dr2/dx with dr2/dx1_d = 2 exp(theta_d) (x1_d - x2_d).
    The reference tests check it by finite differences
    (tests/test_GaussianProcess.py:1010-1015); so does tests/test_oracle_golden.py.
    """
    scale = np.exp(np.asarray(corr_raw, dtype=np.float64))
    diff = x1[:, np.newaxis, :] - x2[np.newaxis, :, :]
    if kernel == PRODMAT52:
        r2 = calc_r2_product(x1, x2, corr_raw)
        Kd = calc_K(r2, MAT52)
        dfac = calc_dKdr2(r2, MAT52) * 2. * scale * diff
        out = np.empty((r2.shape[-1],) + r2.shape[:2])
        for p in range(r2.shape[-1]):
            f = Kd.copy()
            f[:, :, p] = dfac[:, :, p]
            out[p] = np.prod(f, axis=-1)
        return out
    if kernel in (UNISQEXP, UNIMAT52):
        dr2dx = np.transpose(2. * scale[0] * diff, (2, 0, 1))
        return calc_dKdr2(calc_r2_uniform(x1, x2, corr_raw), _BASE[kernel]) * dr2dx
    dr2dx = np.transpose(2. * scale * diff, (2, 0, 1))
    return calc_dKdr2(calc_r2(x1, x2, corr_raw), kernel) * dr2dx


# ----------------------------------------------------------------------------
# linalg/cholesky.py
# ----------------------------------------------------------------------------

def _check_cholesky_inputs(A):
    """linalg/cholesky.py:196-222."""
    A = np.array(A)
    assert A.ndim == 2 and A.shape[0] == A.shape[1], "A must have shape (n,n)"
    np.testing.assert_allclose(A.T, A)
    if np.any(np.diag(A) <= 0.0):
        raise linalg.LinAlgError("not pd: non-positive diagonal elements")
    return A


def fixed_cholesky(A):
    """linalg/cholesky.py:225-231."""
    A = _check_cholesky_inputs(A)
    return linalg.cholesky(A, lower=True)


def jit_cholesky(A, maxtries=5):
    """linalg/cholesky.py:234-281.  Plain dpotrf first; on failure add
    jitter = 1e-6 * mean(diag A) * 10^k, k = 0 .. maxtries-1."""
    A = _check_cholesky_inputs(A)
    assert int(maxtries) > 0
    A = np.ascontiguousarray(A)
    L, info = lapack.dpotrf(A, lower=1)
    if info == 0:
        return L, 0.0
    jitter = np.diag(A).mean() * 1e-6
    num_tries = 1
    while num_tries <= maxtries and np.isfinite(jitter):
        try:
            L = linalg.cholesky(A + np.eye(A.shape[0]) * jitter, lower=True)
            return L, jitter
        except Exception:
            jitter *= 10
        finally:
            num_tries += 1
    raise linalg.LinAlgError("not positive definite, even with jitter.")


class PivotFactor(object):
    """ChoInvPivot, linalg/cholesky.py:82-165: the factor of A[P][:, P] and the pivot order P."""

    def __init__(self, L, P):
        self.L, self.P = L, np.asarray(P)
        self.shape = L.shape

    def inverse_order(self):
        """_pivot_transpose, linalg/cholesky.py:330-357."""
        Pt = np.empty_like(self.P)
        Pt[self.P] = np.arange(len(self.P))
        return Pt


def pivot_cholesky(A):
    """linalg/cholesky.py:284-327: LAPACK dpstrf (lower); the rows the factorisation skipped as collinear keep
    whatever dpstrf left in them below the diagonal, and their diagonal entries are replaced by
    L[rank-1, rank-1] / ((rank+1)(rank+2)...(i+1)), so that logdet and the triangular solves stay defined."""
    A = _check_cholesky_inputs(A)
    A = np.ascontiguousarray(A)
    L, P, rank, info = lapack.dpstrf(A, lower=1)
    L = np.tril(L)
    if info < 0:
        raise linalg.LinAlgError("Illegal value in covariance matrix")
    n = A.shape[0]
    idx = np.arange(rank, n)
    divs = np.cumprod(np.arange(rank + 1, n + 1, dtype=np.float64))
    L[idx, idx] = L[rank - 1, rank - 1] / divs
    return L, P - 1, rank


def cholesky_factor(A, nugget, nugget_type):
    """linalg/cholesky.py:168-193."""
    if nugget_type == "adaptive":
        L, nugget = jit_cholesky(A)
    elif nugget_type == "pivot":
        L, P, _ = pivot_cholesky(A)
        L = PivotFactor(L, P)
    elif nugget_type in ("fit", "fixed"):
        A = A + nugget * np.eye(A.shape[0])
        L = fixed_cholesky(A)
    else:
        raise ValueError("Bad value for nugget_type in cholesky_factor")
    return L, nugget


def cho_solve_L(L, b):
    """ChoInv.solve, linalg/cholesky.py:22-42; ChoInvPivot.solve :105-133."""
    if isinstance(L, PivotFactor):
        if L.shape == (1, 1):
            return b / L.L[0, 0] ** 2
        return cho_solve((L.L, True), b[L.P])[L.inverse_order()]
    if L.shape == (1, 1):
        return b / L[0, 0] ** 2
    return cho_solve((L, True), b)


def solve_L(L, b):
    """ChoInv.solve_L, linalg/cholesky.py:44-65: L^-1 b; ChoInvPivot.solve_L :135-165: L^-1 b[P]."""
    if isinstance(L, PivotFactor):
        return linalg.solve_triangular(L.L, b[L.P], lower=True)
    return linalg.solve_triangular(L, b, lower=True)


def logdet_L(L):
    """ChoInv.logdet, linalg/cholesky.py:67-79."""
    if isinstance(L, PivotFactor):
        L = L.L
    return 2.0 * np.sum(np.log(np.diag(L)))


def logdet_deriv(L, dKdtheta):
    """linalg/linalg_utils.py:170-198: tr(K^-1 dK_p) for each p, computed as the
    trace of cho_solve on the (n, n, P) stacked right-hand side."""
    dK = np.transpose(dKdtheta, (1, 2, 0))
    n, _, P = dK.shape
    sol = cho_solve_L(L, dK.reshape(n, n * P)).reshape(n, n, P)
    return np.trace(sol, axis1=0, axis2=1)


# ----------------------------------------------------------------------------
# Priors.py (log-density pieces entering the posterior)
# ----------------------------------------------------------------------------

class Prior(object):
    """(kind, shape, scale) prior on a *scaled* hyper-parameter.

    kind in {"weak", "invgamma", "gamma", "lognormal"}:
    WeakPrior Priors.py:578-649; LogNormalPrior :842-901; GammaPrior :903-971;
    InvGammaPrior :973-1128.
    """

    def __init__(self, kind="weak", shape=0., scale=0.):
        self.kind, self.shape, self.scale = kind.lower(), float(shape), float(scale)

    def logp(self, x):
        a, b = self.shape, self.scale
        if self.kind == "weak":
            return 0.
        if self.kind == "invgamma":
            return a * np.log(b) - gammaln(a) - (a + 1.) * np.log(x) - b / x
        if self.kind == "gamma":
            return -a * np.log(b) - gammaln(a) + (a - 1.) * np.log(x) - x / b
        if self.kind == "lognormal":
            return (-0.5 * (np.log(x / b) / a) ** 2 - 0.5 * np.log(2. * np.pi)
                    - np.log(x) - np.log(a))
        raise ValueError(self.kind)

    def dlogpdx(self, x):
        a, b = self.shape, self.scale
        if self.kind == "weak":
            return 0.
        if self.kind == "invgamma":
            return -(a + 1.) / x + b / x ** 2
        if self.kind == "gamma":
            return (a - 1.) / x - 1. / b
        if self.kind == "lognormal":
            return -np.log(x / b) / a ** 2 / x - 1. / x
        raise ValueError(self.kind)


class GPPriorsRef(object):
    """GPPriors.logp / dlogpdtheta, Priors.py:291-354.

    ``corr`` acts on l_d = exp(-theta_d/2) with dl/dtheta = -l/2
    (GPParams.py:35-67); ``cov`` and ``nugget`` act on exp(theta) with
    d/dtheta = exp(theta) (GPParams.py:115-147)."""

    def __init__(self, n_corr, nugget_type, corr=None, cov=None, nugget=None):
        self.n_corr = n_corr
        self.nugget_type = nugget_type
        self.corr = list(corr) if corr is not None else [Prior() for _ in range(n_corr)]
        self.cov = cov if cov is not None else Prior()
        self.nugget = nugget if nugget is not None else Prior()

    def logp(self, theta):
        D = self.n_corr
        lp = 0.
        for p, th in zip(self.corr, theta[:D]):
            lp += p.logp(np.exp(-0.5 * th))
        lp += self.cov.logp(np.exp(theta[D]))
        if self.nugget_type == "fit":
            lp += self.nugget.logp(np.exp(theta[D + 1]))
        return lp

    def dlogpdtheta(self, theta):
        D = self.n_corr
        out = []
        for p, th in zip(self.corr, theta[:D]):
            l = np.exp(-0.5 * th)
            out.append(float(p.dlogpdx(l) * (-0.5 * l)))
        s2 = np.exp(theta[D])
        out.append(float(self.cov.dlogpdx(s2) * s2))
        if self.nugget_type == "fit":
            eta = np.exp(theta[D + 1])
            out.append(float(self.nugget.dlogpdx(eta) * eta))
        return np.array(out)


# ----------------------------------------------------------------------------
# GaussianProcess.py (mean=None branch)
# ----------------------------------------------------------------------------

class GPRef(object):
    """Zero-mean restatement of ``GaussianProcess``.

    nugget: "adaptive", "fit" or a non-negative float (GPParams.py:165-199).
    theta layout: [corr_raw (D) | log sigma^2 | log nugget (fit only)]
    (GPParams.py:215-545)."""

    def __init__(self, inputs, targets, kernel=SQEXP, nugget="adaptive", priors=None,
                 chunk_rows=None):
        self.X = np.ascontiguousarray(np.asarray(inputs, dtype=np.float64))
        if self.X.ndim == 1:
            self.X = self.X.reshape(-1, 1)
        self.t = np.asarray(targets, dtype=np.float64)
        assert self.t.ndim == 1 and self.t.shape[0] == self.X.shape[0]
        self.n, self.D = self.X.shape
        self.kernel = kernel
        if isinstance(nugget, str):
            assert nugget in ("adaptive", "fit", "pivot")
            self.nugget_type, self.nugget = nugget, None
        else:
            assert float(nugget) >= 0.
            self.nugget_type, self.nugget = "fixed", float(nugget)
        self.nc = n_corr_of(kernel, self.D)            # number of correlation parameters (1 for the uniform kernels)
        self.n_params = self.nc + 1 + int(self.nugget_type == "fit")
        self.priors = priors if priors is not None else GPPriorsRef(self.nc, self.nugget_type)
        self.theta = None
        self.chunk_rows = chunk_rows
        self.L = self.Kinv_t = self.current_logpost = None

    # -- helpers ------------------------------------------------------------
    def _r2(self, a, b, corr_raw):
        if self.chunk_rows:
            return calc_r2_chunked(a, b, corr_raw, self.chunk_rows)
        return calc_r2(a, b, corr_raw)

    def get_cov_matrix(self, other):
        """GaussianProcess.py:517-543: sigma^2 k(X, other), shape (n, m)."""
        D = self.nc
        if self.kernel not in (SQEXP, MAT52):
            return np.exp(self.theta[D]) * kernel_f(self.X, other, self.theta[:D], self.kernel)
        return np.exp(self.theta[D]) * calc_K(self._r2(self.X, other, self.theta[:D]), self.kernel)

    def get_K_matrix(self):
        """GaussianProcess.py:545-558 (no nugget)."""
        return self.get_cov_matrix(self.X)

    def _refit(self, theta):
        """GaussianProcess.py:606-627."""
        return self.theta is None or not np.allclose(theta, self.theta, rtol=1.e-10, atol=1.e-15)

    # -- fit / objective ----------------------------------------------------
    def fit(self, theta):
        """GaussianProcess.py:629-685 with H = (n x 0)."""
        theta = np.array(theta, dtype=np.float64)
        assert theta.shape == (self.n_params,), "bad shape for hyperparameters"
        self.theta = theta
        if self.nugget_type == "fit":
            self.nugget = float(np.exp(theta[-1]))
        K = self.get_K_matrix()
        self.L, newnugget = cholesky_factor(K, self.nugget, self.nugget_type)
        if self.nugget_type == "adaptive":
            self.nugget = float(newnugget)
        self.Kinv_t = cho_solve_L(self.L, self.t)
        self.current_logpost = 0.5 * (np.dot(self.t, self.Kinv_t) + logdet_L(self.L)
                                      + self.n * np.log(2. * np.pi))
        self.current_logpost -= self.priors.logp(theta)
        return self.current_logpost

    def logposterior(self, theta):
        """GaussianProcess.py:688-709."""
        if self._refit(theta):
            self.fit(theta)
        return self.current_logpost

    def logpost_deriv(self, theta):
        """GaussianProcess.py:711-782 with w = 0, A = 0x0."""
        if self._refit(theta):
            self.fit(theta)
        D = self.nc
        partials = np.zeros(self.n_params)
        dKdtheta = np.exp(self.theta[D]) * kernel_deriv(self.X, self.X, self.theta[:D], self.kernel)
        a = self.Kinv_t
        partials[:D] = 0.5 * (-np.dot(a, np.dot(dKdtheta, a).T) + logdet_deriv(self.L, dKdtheta))
        dKdcov = self.get_K_matrix().reshape(1, self.n, self.n)
        partials[D] = 0.5 * (-np.dot(a, np.dot(dKdcov[0], a)) + logdet_deriv(self.L, dKdcov)[0])
        if self.nugget_type == "fit":
            eye = np.eye(self.n).reshape(1, self.n, self.n)
            partials[-1] = 0.5 * self.nugget * (-np.dot(a, a) + logdet_deriv(self.L, eye)[0])
        partials -= self.priors.dlogpdtheta(self.theta)
        return partials

    def logpost_deriv_chunked(self, theta, chunk_rows=128):
        """GaussianProcess.py:711-782 at sizes where the (D, n, n) ``kernel_deriv`` tensor and the (n, n D) right-hand
        side of ``logdet_deriv`` (linalg/linalg_utils.py:170-198) do not fit (C4: 4 GB + 4 GB, C5: 16 GB each).
        DEVIATION, flagged as SURVEY.md section 8d allows for the memory guard: tr(K^-1 dK_p) is accumulated as
        sum_ij [K^-1]_ij [dK_p]_ij over row blocks of dK_p, with K^-1 from LAPACK (cho_solve on the identity), instead of
        as the trace of cho_solve on the stacked planes -- the same quantity, the per-entry arithmetic of dK_p is the
        reference's (``kernel_deriv`` on a row block).  tests/test_oracle_golden.py pins it to ``logpost_deriv``."""
        if self._refit(theta):
            self.fit(theta)
        D, n = self.nc, self.n
        Kinv = cho_solve_L(self.L, np.eye(n))
        a = self.Kinv_t
        sig2 = np.exp(self.theta[D])
        tr = np.zeros(D + 1)
        quad = np.zeros(D + 1)
        for lo in range(0, n, chunk_rows):
            hi = min(lo + chunk_rows, n)
            dK = sig2 * kernel_deriv(self.X[lo:hi], self.X, self.theta[:D], self.kernel)          # (D, rows, n)
            Kc = self.get_cov_matrix(self.X[lo:hi]).T                                              # (rows, n), no nugget
            W = Kinv[lo:hi]
            for p in range(D):
                tr[p] += np.sum(W * dK[p])
                quad[p] += np.dot(a[lo:hi], np.dot(dK[p], a))
            tr[D] += np.sum(W * Kc)
            quad[D] += np.dot(a[lo:hi], np.dot(Kc, a))
        partials = np.zeros(self.n_params)
        partials[:D + 1] = 0.5 * (tr - quad)
        if self.nugget_type == "fit":
            partials[-1] = 0.5 * self.nugget * (np.trace(Kinv) - np.dot(a, a))
        partials -= self.priors.dlogpdtheta(self.theta)
        return partials

    # -- predict ------------------------------------------------------------
    def predict(self, testing, unc=True, deriv=False, include_nugget=True, full_cov=False):
        """GaussianProcess.py:818-927 (R = 0; ``full_cov`` :899-911).  ``deriv=True``
        returns the analytic input-derivative of the mean, which is what
        DenseGP_GPU::predict_deriv (densegp_gpu.hpp:411-448) returns."""
        if self.theta is None:
            raise ValueError("hyperparameters have not been fit for this Gaussian Process")
        testing = np.asarray(testing, dtype=np.float64)
        if testing.ndim == 1:
            testing = testing.reshape(-1, 1) if self.D == 1 else testing.reshape(1, -1)
        assert testing.shape[1] == self.D
        Ktest = self.get_cov_matrix(testing)
        mu = np.dot(Ktest.T, self.Kinv_t)
        var = None
        if unc:
            Kinv_Ktest = cho_solve_L(self.L, Ktest)
            sigma_2 = np.exp(self.theta[self.nc])
            if full_cov:
                Kss = sigma_2 * kernel_f(testing, testing, self.theta[:self.nc], self.kernel)
                if include_nugget and self.nugget_type != "pivot":       # GaussianProcess.py:904,915
                    Kss = Kss + np.eye(testing.shape[0]) * self.nugget
                Linv_Ktest = solve_L(self.L, Ktest)
                var = Kss - np.dot(Linv_Ktest.T, Linv_Ktest)
            else:
                if include_nugget and self.nugget_type != "pivot":       # GaussianProcess.py:904,915
                    sigma_2 = sigma_2 + self.nugget
                var = np.maximum(sigma_2 - np.sum(Ktest * Kinv_Ktest, axis=0), 0.)
        d = None
        if deriv:
            dk = np.exp(self.theta[self.nc]) * kernel_inputderiv(testing, self.X, self.theta[:self.nc],
                                                                self.kernel)
            d = np.einsum("dmj,j->md", dk, self.Kinv_t)
        return mu, var, d


# ----------------------------------------------------------------------------
# GaussianProcess.py with a mean function (analytic mean, weak mean priors) -- SURVEY.md section 8f row 1
# ----------------------------------------------------------------------------

def design_matrix(X, terms, intercept=True):
    """Design matrix H (n, q) of the polynomial mean formulae the GPU wrapper understands:
    an intercept column (patsy adds it for formulas like "x[0]", GaussianProcess.py:485-515) followed by
    x[dim]^power columns for ``terms = [(dim, power), ...]``."""
    X = np.asarray(X, dtype=np.float64)
    cols = [np.ones(X.shape[0])] if intercept else []
    for dim, power in terms:
        cols.append(X[:, dim] ** power)
    return np.stack(cols, axis=1) if cols else np.zeros((X.shape[0], 0))


class GPRefMean(GPRef):
    """``GaussianProcess`` with a linear-in-parameters mean whose coefficients are integrated out
    analytically under weak (improper flat) mean priors: the general branch of
    GaussianProcess.fit / logpost_deriv / predict (GaussianProcess.py:657-685, 745-778, 885-920) with
    B^-1 = 0, b = 0 (MeanPriors weak: Priors.py:430-470, inv_cov() = 0, logdet_cov() = 0)."""

    def __init__(self, inputs, targets, terms, intercept=True, mean_prior=None, **kw):
        """``mean_prior``: None (weak) or ``(b, cov)`` -- MeanPriors(mean=b, cov=cov), Priors.py:423-581, cov a
        scalar / vector of variances / covariance matrix."""
        super().__init__(inputs, targets, **kw)
        self.terms, self.intercept = list(terms), intercept
        self.H = design_matrix(self.X, self.terms, intercept)
        self.q = self.H.shape[1]
        if mean_prior is None:
            self.b = self.Bcov = None
        else:
            self.b = np.reshape(np.array(mean_prior[0], dtype=np.float64), (-1,))
            self.Bcov = np.array(mean_prior[1], dtype=np.float64)
            assert len(self.b) == self.q

    # MeanPriors.inv_cov / inv_cov_b / logdet_cov / dm_dot_b, Priors.py:493-579
    def _inv_cov(self):
        if self.b is None:
            return 0.
        if self.Bcov.ndim < 2:
            return np.diag(np.broadcast_to(1. / self.Bcov, (self.q,)))
        return cho_solve(cho_factor(self.Bcov), np.eye(self.q))

    def _inv_cov_b(self):
        if self.b is None:
            return 0.
        if self.Bcov.ndim < 2:
            return self.b / self.Bcov
        return cho_solve(cho_factor(self.Bcov), self.b)

    def _logdet_cov(self):
        if self.b is None:
            return 0.
        if self.Bcov.ndim < 2:
            return float(np.sum(np.log(np.broadcast_to(self.Bcov, (self.q,)))))
        return 2. * float(np.sum(np.log(np.diag(cho_factor(self.Bcov)[0]))))

    def fit(self, theta):
        theta = np.array(theta, dtype=np.float64)
        assert theta.shape == (self.n_params,), "bad shape for hyperparameters"
        self.theta = theta
        if self.nugget_type == "fit":
            self.nugget = float(np.exp(theta[-1]))
        H = self.H
        m = np.zeros(self.n) if self.b is None else np.dot(H, self.b)          # priors.mean.dm_dot_b, GaussianProcess.py:657
        K = self.get_K_matrix()
        self.L, newnugget = cholesky_factor(K, self.nugget, self.nugget_type)
        if self.nugget_type == "adaptive":
            self.nugget = float(newnugget)
        # calc_Ainv, linalg_utils.py:5-40:  A = H^T K^-1 H + B^-1
        self.Kinv_H = cho_solve_L(self.L, H)
        A = np.dot(H.T, self.Kinv_H) + self._inv_cov()
        self.LA = fixed_cholesky(A)
        self.Kinv_t = cho_solve_L(self.L, self.t - m)
        H_Kinv_t = np.dot(H.T, self.Kinv_t)
        # calc_mean_params, linalg_utils.py:88-121
        self.beta = cho_solve_L(self.LA, H_Kinv_t + self._inv_cov_b())
        self.Kinv_t_mean = cho_solve_L(self.L, self.t - np.dot(H, self.beta))
        n_coeff = self.n - self.q if self.b is None else self.n             # GaussianProcess.py:674-677
        self.current_logpost = 0.5 * (np.dot(self.t - m, self.Kinv_t) - np.dot(H_Kinv_t, cho_solve_L(self.LA, H_Kinv_t))
                                      + logdet_L(self.L) + logdet_L(self.LA) + self._logdet_cov() + n_coeff * np.log(2. * np.pi))
        self.current_logpost -= self.priors.logp(theta)
        return self.current_logpost

    def logpost_deriv(self, theta):
        if self._refit(theta):
            self.fit(theta)
        D, H = self.nc, self.H
        partials = np.zeros(self.n_params)
        a = self.Kinv_t
        u = cho_solve_L(self.L, np.dot(H, cho_solve_L(self.LA, np.dot(H.T, a))))      # Kinv_H_Ainv_H_Kinv_t, :747-749

        def dA(dK):      # calc_A_deriv, linalg_utils.py:42-86
            return -np.einsum("ic,pij,jd->pcd", self.Kinv_H, dK, self.Kinv_H)

        def quad(dK):
            return (-np.einsum("i,pij,j->p", a, dK, a) + 2. * np.einsum("i,pij,j->p", a, dK, u)
                    - np.einsum("i,pij,j->p", u, dK, u))

        dKdtheta = np.exp(self.theta[D]) * kernel_deriv(self.X, self.X, self.theta[:D], self.kernel)
        partials[:D] = 0.5 * (quad(dKdtheta) + logdet_deriv(self.L, dKdtheta) + logdet_deriv(self.LA, dA(dKdtheta)))
        dKdcov = self.get_K_matrix().reshape(1, self.n, self.n)
        partials[D] = 0.5 * (quad(dKdcov) + logdet_deriv(self.L, dKdcov) + logdet_deriv(self.LA, dA(dKdcov)))[0]
        if self.nugget_type == "fit":
            eye = np.eye(self.n).reshape(1, self.n, self.n)
            partials[-1] = 0.5 * self.nugget * (quad(eye) + logdet_deriv(self.L, eye) + logdet_deriv(self.LA, dA(eye)))[0]
        partials -= self.priors.dlogpdtheta(self.theta)
        return partials

    def predict(self, testing, unc=True, deriv=False, include_nugget=True, full_cov=False):
        if self.theta is None:
            raise ValueError("hyperparameters have not been fit for this Gaussian Process")
        testing = np.asarray(testing, dtype=np.float64)
        if testing.ndim == 1:
            testing = testing.reshape(-1, 1) if self.D == 1 else testing.reshape(1, -1)
        Hs = design_matrix(testing, self.terms, self.intercept)
        Ktest = self.get_cov_matrix(testing)
        mu = np.dot(Hs, self.beta) + np.dot(Ktest.T, self.Kinv_t_mean)                  # :888-891
        var = None
        if unc:
            Kinv_Ktest = cho_solve_L(self.L, Ktest)
            Rm = Hs.T - np.dot(self.H.T, Kinv_Ktest)                                     # calc_R, linalg_utils.py:123-168
            sigma_2 = np.exp(self.theta[self.nc])
            if full_cov:                                                                 # :899-911
                Kss = sigma_2 * kernel_f(testing, testing, self.theta[:self.nc], self.kernel)
                if include_nugget and self.nugget_type != "pivot":       # GaussianProcess.py:904,915
                    Kss = Kss + np.eye(testing.shape[0]) * self.nugget
                Linv_Ktest = solve_L(self.L, Ktest)
                LAinv_R = solve_L(self.LA, Rm)
                var = Kss - np.dot(Linv_Ktest.T, Linv_Ktest) + np.dot(LAinv_R.T, LAinv_R)
                return mu, var, None
            if include_nugget and self.nugget_type != "pivot":       # GaussianProcess.py:904,915
                sigma_2 = sigma_2 + self.nugget
            var = np.maximum(sigma_2 - np.sum(Ktest * Kinv_Ktest, axis=0) + np.sum(Rm * cho_solve_L(self.LA, Rm), axis=0), 0.)
        return mu, var, None


# ----------------------------------------------------------------------------
# fitting.py
# ----------------------------------------------------------------------------

def fit_GP_MAP_ref(gp, n_tries=15, theta0=None, method="L-BFGS-B", sampler=None, **kwargs):
    """fitting.py:219-266 (_fit_single_GP_MAP): multi-start minimisation of
    the negative log-posterior, keep the best.  ``sampler()`` returns a start
    point (the reference draws it from the priors, Priors.py:394-418)."""
    from scipy.optimize import minimize
    if sampler is None:
        rng = np.random.default_rng(0)
        sampler = lambda: 5. * (rng.random(gp.n_params) - 0.5)  # WeakPrior.sample, Priors.py:636-649
    old = np.seterr(divide="raise", over="raise", invalid="raise")
    vals, thetas = [], []
    try:
        for i in range(int(n_tries)):
            theta = np.array(theta0) if (i == 0 and theta0 is not None) else sampler()
            try:
                res = minimize(gp.logposterior, theta, method=method, jac=gp.logpost_deriv,
                               options=kwargs)
                vals.append(res["fun"])
                thetas.append(res["x"])
            except (linalg.LinAlgError, FloatingPointError):
                pass
    finally:
        np.seterr(**old)
    if not vals:
        gp.theta = None
        return gp
    gp.fit(thetas[int(np.argmin(vals))])
    return gp


# ----------------------------------------------------------------------------
# consumers of predict
# ----------------------------------------------------------------------------

def implausibility_ref(obs, obs_var, mean, var, discrepancy=0., rank=1):
    """HistoryMatching.get_implausibility, HistoryMatching.py:236-276, for given predictions.
    obs / obs_var: (n_obs,), mean / var: (n_obs, m) (or (m,) for one output), discrepancy scalar or (n_obs,)."""
    obs, obs_var = np.atleast_1d(np.asarray(obs, dtype=np.float64)), np.atleast_1d(np.asarray(obs_var, dtype=np.float64))
    mean, var = np.atleast_2d(mean), np.atleast_2d(var)
    discrepancy = np.atleast_1d(discrepancy)
    n_obs = len(obs)
    if n_obs == 1:
        rank = 0
    assert 0 <= rank < n_obs
    Vs = np.zeros(mean.shape)
    Vs += var
    Vs += discrepancy[:, np.newaxis]
    Vs += obs_var[:, np.newaxis]
    I = np.abs(obs[:, np.newaxis] - mean) / np.sqrt(Vs)
    return np.partition(I, n_obs - rank - 1, axis=0)[n_obs - rank - 1]


def mice_fast_predict_ref(gp, index):
    """MICEFastGP.fast_predict, SequentialDesign.py:705-747: predictive variance at training input ``index``
    of the fitted GPRef ``gp`` when that point is excluded, via the Woodbury downdate of the full inverse."""
    n, D = gp.n, gp.nc
    keep = np.arange(n) != index
    sigma_2 = np.exp(gp.theta[D]) + gp.nugget
    Ktest = np.exp(gp.theta[D]) * kernel_f(gp.X[keep], gp.X[index:index + 1], gp.theta[:D], gp.kernel)
    invQ = np.linalg.solve(gp.L.T, np.linalg.solve(gp.L, np.eye(n)))
    invQ_mod = invQ[keep][:, keep] - np.outer(invQ[keep, index], invQ[keep, index]) / invQ[index, index]
    return np.maximum(sigma_2 - np.sum(Ktest * np.dot(invQ_mod, Ktest), axis=0), 0.)


def mice_criterion_ref(gp, candidates, nugget_s=1.):
    """MICEDesign._MICE_criterion for every candidate, SequentialDesign.py:884-911 and :941-964 (the candidate GP
    takes the base GP's correlation lengths and covariance and the nugget base_nugget * nugget_s)."""
    candidates = np.asarray(candidates, dtype=np.float64)
    fast = GPRef(candidates, np.ones(len(candidates)), kernel=gp.kernel, nugget=float(gp.nugget * nugget_s))
    fast.fit(gp.theta[:gp.nc + 1])
    out = np.zeros(len(candidates))
    for c in range(len(candidates)):
        unc1 = gp.predict(candidates[c], unc=True)[1]
        out[c] = unc1[0] / mice_fast_predict_ref(fast, c)[0]
    return out


# ----------------------------------------------------------------------------
# validation.py: errors on a validation set from the predictions
# ----------------------------------------------------------------------------

def standard_errors_ref(target, mean, var):
    """StandardErrors.__call__, validation.py:367-398: errors in order of decreasing predictive variance."""
    P = np.argsort(var)[::-1]
    return ((mean - target) / np.sqrt(var))[P], P


def pivoted_errors_ref(target, mean, cov):
    """PivotErrors.__call__, validation.py:401-441: cholesky_factor(cov, 0., "pivot") then ChoInvPivot.solve_L."""
    L, P, _ = pivot_cholesky(cov)
    return linalg.solve_triangular(L, (mean - target)[P], lower=True), P


def mahalanobis_ref(target, mean, cov, n_train=None, n_mean=0, scaled=False):
    """mahalanobis, validation.py:8-95 (one emulator): sum of squared pivoted errors; scaled by the mean and standard
    deviation of F(n_valid, n_train - n_mean - 2) with scale n_valid (generate_mahal_dist, :98-135)."""
    from scipy.stats import f
    err, _ = pivoted_errors_ref(target, mean, cov)
    M = float(np.sum(err ** 2))
    if scaled:
        nv = len(target)
        mu, var = f(dfn=nv, dfd=n_train - n_mean - 2, scale=nv).stats()
        M = (M - mu) / np.sqrt(var)
    return M
