"""
Validation diagnostics on top of the device path -- the consumers of ``predict(full_cov=True)`` of
mogp_emulator/validation.py:8-482, for ``GaussianProcessGPU`` and ``MultiOutputGP_GPU``:

* ``standard_errors``  (y_pred - y_valid) / sqrt(var), ordered by decreasing predictive variance (validation.py:240-293, 367-398)
* ``pivoted_errors``   L^-1 (y_pred - y_valid)[P] with the pivoted Cholesky factor of the predictive covariance, i.e. the
                       errors de-correlated in order of decreasing conditional variance (validation.py:296-338, 401-441)
* ``mahalanobis``      sum of the squared pivoted errors, optionally scaled by the mean / standard deviation of its
                       Fisher-Snedecor reference distribution (validation.py:8-95); ``generate_mahal_dist`` (validation.py:98-135)

The predictive mean / variance / full covariance come from the batched device prediction and the pivoted factorisation of
each (n_valid x n_valid) covariance from the device routine behind ``nugget="pivot"`` (``LibGPGPU.pivot_cholesky``); what is
left for the host is one triangular solve with n_valid right-hand-side entries.  Same function names, argument meaning,
return shapes and error behaviour as the reference module.
"""
import numpy as np
from scipy.linalg import solve_triangular
from scipy.stats import f as _fisher_snedecor

from . import LibGPGPU
from .GaussianProcessGPU import GaussianProcessGPU
from .MultiOutputGP_GPU import MultiOutputGP_GPU


class Errors(object):
    "base class of the error definitions (validation.py:341-350)"
    full_cov = False

    def __call__(self, target, mean, cov):
        raise NotImplementedError


class StandardErrors(Errors):
    full_cov = False

    def __call__(self, target, mean, cov):
        P = np.argsort(cov)[::-1]
        return ((mean - target) / np.sqrt(cov))[P], P


class PivotErrors(Errors):
    full_cov = True

    def __call__(self, target, mean, cov):
        # cholesky_factor(cov, 0., "pivot") + ChoInvPivot.solve_L of the reference, factorised on the device
        L, P, _ = LibGPGPU.pivot_cholesky(cov)
        return solve_triangular(L, (mean - target)[P], lower=True), P


def _is_single(gp):
    return isinstance(gp, GaussianProcessGPU)


def _process_inputs(gp, inputs):
    inputs = np.array(inputs, dtype=np.float64)
    if inputs.ndim == 1:
        inputs = inputs.reshape(-1, 1) if gp.D == 1 else inputs.reshape(1, -1)
    return inputs


def _check_valid_data(gp, valid_inputs, valid_targets):
    assert isinstance(gp, (GaussianProcessGPU, MultiOutputGP_GPU)), "Must provide a GP to validate"
    valid_inputs = _process_inputs(gp, valid_inputs)
    valid_targets = np.array(valid_targets)
    if _is_single(gp):
        assert valid_targets.ndim == 1, "Targets for a GP must be a 1D array"
        assert valid_targets.shape[0] == valid_inputs.shape[0], "Bad length for validation targets"
    else:
        assert valid_targets.ndim == 2, "Targets for a MultiOutputGP must be a 2D array"
        assert valid_targets.shape[1] == valid_inputs.shape[0], "Bad shape for validation targets"
    return valid_inputs, valid_targets


def _n_mean(em):
    "number of mean-function coefficients of an emulator, whether they live in theta or are integrated out"
    native = em._densegp_gpu
    return max(int(native.get_theta().get_n_mean()), int(native.get_beta().size))


def compute_errors(gp, valid_inputs, valid_targets, method):
    if isinstance(method, str):
        # (the reference compares the unbound ``method.lower`` and therefore rejects every string, validation.py:210-216;
        # the names it documents are accepted here)
        key = method.lower()
        if key in ("standard", "standarderrors"):
            method = StandardErrors()
        elif key in ("pivot", "pivoterrors"):
            method = PivotErrors()
        else:
            raise ValueError("Bad value for error method in compute_errors")
    assert issubclass(type(method), Errors), "method must be a subclass of Errors"
    valid_inputs, valid_targets = _check_valid_data(gp, valid_inputs, valid_targets)
    mean, cov, _ = gp.predict(valid_inputs, deriv=False, full_cov=method.full_cov)
    if _is_single(gp):
        return method(valid_targets, mean, cov)
    return [method(t, m, c) for t, m, c in zip(valid_targets, mean, cov)]


def standard_errors(gp, valid_inputs, valid_targets):
    return compute_errors(gp, valid_inputs, valid_targets, method=StandardErrors())


def pivoted_errors(gp, valid_inputs, valid_targets):
    return compute_errors(gp, valid_inputs, valid_targets, method=PivotErrors())


def generate_mahal_dist(gp, valid_inputs):
    if _is_single(gp):
        emulators = [gp]
    elif isinstance(gp, MultiOutputGP_GPU):
        emulators = gp.emulators
    else:
        raise TypeError("Provided GP is not a GaussianProcess or MultiOutputGP")
    n_valid = len(_process_inputs(gp, valid_inputs))
    dists = [_fisher_snedecor(dfn=n_valid, dfd=em.n - _n_mean(em) - 2, scale=n_valid) for em in emulators]
    return dists[0] if len(dists) == 1 else dists


def mahalanobis(gp, valid_inputs, valid_targets, scaled=False):
    pivot_errors = pivoted_errors(gp, valid_inputs, valid_targets)
    if _is_single(gp):
        errors = pivot_errors[0]
    else:
        errors = np.array([err[0] for err in pivot_errors])
    M = np.sum(errors ** 2, axis=-1)
    if scaled:
        dists = generate_mahal_dist(gp, valid_inputs)
        single = _is_single(gp) or not isinstance(dists, list)
        M_iter, d_iter = ([M], [dists]) if single else (M, dists)
        out = []
        for M_val, dist in zip(M_iter, d_iter):
            mean, var = dist.stats()
            out.append((M_val - mean) / np.sqrt(var))
        M = np.array(out)
        if _is_single(gp):
            M = M.squeeze(axis=0)
    return M
