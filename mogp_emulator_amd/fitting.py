"""
``fit_GP_MAP`` for the GPU classes -- mirror of the GPU branches of mogp_emulator/fitting.py:16-217.
The optimisation itself is one native call (``LibGPGPU.fit_GP_MAP``): multi-start L-BFGS, every (emulator, start) run
advancing independently through batched device evaluations (a slot pool: a run that ends hands its slot to the next one).
"""
import numpy as np

from . import LibGPGPU
from .GaussianProcessGPU import GaussianProcessGPU
from .MultiOutputGP_GPU import MultiOutputGP_GPU

_GP_KWARGS = ("mean", "kernel", "priors", "nugget", "inputdict", "use_patsy")


def _check_common(n_tries, method):
    if method not in ("L-BFGS", "L-BFGS-B"):
        raise NotImplementedError("Unknown method for optimizer - only L-BFGS implemented for GPU")
    n_tries = int(n_tries)
    assert n_tries > 0, "number of attempts must be positive"
    return n_tries


def _fit_single_GPGPU_MAP(gp, n_tries=15, theta0=None, method="L-BFGS-B", **kwargs):
    n_tries = _check_common(n_tries, method)
    theta0 = np.array([]) if theta0 is None or len(theta0) == 0 else np.asarray(theta0, dtype=np.float64)
    LibGPGPU.fit_GP_MAP(gp._densegp_gpu, n_tries, theta0)
    if not gp.theta.data_has_been_set():
        raise RuntimeError("Fitting did not converge")
    return gp


def _fit_MOGPGPU_MAP(gp, n_tries=15, theta0=None, method="L-BFGS-B", **kwargs):
    n_tries = _check_common(n_tries, method)
    theta0 = np.array([]) if theta0 is None or len(theta0) == 0 else np.asarray(theta0, dtype=np.float64)
    LibGPGPU.fit_GP_MAP(gp._mogp_gpu, n_tries, theta0)
    return gp


def fit_GP_MAP(*args, n_tries=15, theta0=None, method="L-BFGS-B", skip_failures=True, refit=False, **kwargs):
    """Fit one GP or a multi-output GP by maximising the posterior.  Accepts either an existing
    ``GaussianProcessGPU`` / ``MultiOutputGP_GPU`` or ``(inputs, targets)`` plus constructor keywords
    (1-D targets -> single GP, 2-D targets -> multi-output)."""
    if len(args) == 1:
        gp = args[0]
    elif len(args) < 2:
        raise TypeError("missing required inputs/targets arrays to GaussianProcess")
    else:
        gp_kwargs = {k: kwargs.pop(k) for k in _GP_KWARGS if k in kwargs}
        try:
            ndim = np.asarray(args[1]).ndim
            cls = GaussianProcessGPU if ndim == 1 else MultiOutputGP_GPU
            gp = cls(*args, **gp_kwargs)
        except AssertionError:
            raise ValueError("Bad values for *args in fit_GP_MAP")
    if not LibGPGPU.gpu_usable():
        raise RuntimeError("fit_GP_MAP: the GPU library or a compatible GPU is unavailable")
    if isinstance(gp, GaussianProcessGPU):
        try:
            gp = _fit_single_GPGPU_MAP(gp, n_tries, theta0, method, **kwargs)
        except RuntimeError as exc:
            if "did not converge" in str(exc):
                raise RuntimeError("GP fitting failed")
            raise
    elif isinstance(gp, MultiOutputGP_GPU):
        gp = _fit_MOGPGPU_MAP(gp, n_tries, theta0, method, **kwargs)
        bad = gp.get_indices_not_fit()
        if len(bad) > 0:
            msg = "Fitting failed for emulators {}".format(bad)
            if skip_failures:
                print(msg)
            else:
                raise RuntimeError(msg)
    else:
        raise TypeError("single arg to fit_GP_MAP must be a GaussianProcessGPU or MultiOutputGP_GPU instance")
    return gp
