"""
``MultiOutputGP_GPU`` -- host-side mirror of mogp_emulator/MultiOutputGP_GPU.py:27-382.

All emulators share the inputs and live in ONE native engine; ``fit``, ``predict`` and
``fit_GP_MAP`` are single batched device passes over every emulator (the reference loops over
per-emulator objects with OpenMP on one stream, multioutputgp_gpu.hpp:156-228).

Fixes relative to the reference wrapper: priors given as a list/dict are
honoured (the reference passes an unknown keyword), ``__call__`` works, test points are not
limited to batch_size / n_emulators.
"""
import numpy as np

from . import LibGPGPU
from .GaussianProcessGPU import (GaussianProcessGPU, PredictResult, _resolve_kernel, _resolve_mean,
                                 apply_mean_priors, create_prior_params)
from .Priors import GPPriors


class MultiOutputGP_GPU(object):
    def __init__(self, inputs, targets, mean=None, kernel="SquaredExponential", priors=None, nugget="adaptive",
                 inputdict={}, use_patsy=True, batch_size=16000, analytic_mean=False):
        if not LibGPGPU.gpu_usable():
            raise RuntimeError("Cannot construct MultiOutputGP_GPU: the GPU library or a compatible GPU is unavailable")
        inputs = np.array(inputs, dtype=np.float64)
        targets = np.array(targets, dtype=np.float64)
        if inputs.ndim == 1:
            inputs = inputs.reshape(-1, 1)
        if targets.ndim == 1:
            targets = targets.reshape(1, -1)
        elif targets.ndim != 2:
            raise ValueError("targets must be either a 1D or 2D array")
        if inputs.ndim != 2:
            raise ValueError("inputs must be either a 1D or 2D array")
        if inputs.shape[0] != targets.shape[1]:
            raise ValueError("the first dimension of inputs must be the same length as the second dimension of "
                             "targets (or first if targets is 1D))")
        if nugget == "adaptive":
            nugtype, nugsize = LibGPGPU.nugget_type(0), 0.
        elif nugget == "fit":
            nugtype, nugsize = LibGPGPU.nugget_type(1), 0.
        elif nugget == "pivot":
            nugtype, nugsize = LibGPGPU.nugget_type(3), 0.
        elif isinstance(nugget, float):
            if nugget < 0.:
                raise ValueError("nugget parameter must be non-negative")
            nugtype, nugsize, nugget = LibGPGPU.nugget_type(2), nugget, "fixed"
        else:
            raise TypeError("nugget parameter must be a string or a non-negative float")
        self._nugget_name = nugget
        ktype, _ = _resolve_kernel(kernel)
        # analytic_mean=True: mean coefficients integrated out with weak priors (the CPU class's
        # treatment, GaussianProcess.py:640-700) rather than optimised inside theta (the reference GPU class)
        self._analytic_mean = bool(analytic_mean)
        self._mogp_gpu = LibGPGPU.MultiOutputGP_GPU(inputs, targets, batch_size, _resolve_mean(mean), ktype, nugtype, nugsize,
                                                    analytic_mean=bool(analytic_mean))

        if isinstance(priors, (GPPriors, dict)) or priors is None:
            priorslist = [priors] * self.n_emulators
        else:
            priorslist = list(priors)
            assert len(priorslist) == self.n_emulators, "Bad length for list provided for priors to MultiOutputGP"
        self._set_priors(priorslist, nugget)

    def _set_priors(self, priorslist, nugget_type):
        default = None
        for i, pr in enumerate(priorslist):
            if pr:
                params = create_prior_params(newpriors=pr)
            else:
                if default is None:      # every emulator shares the inputs, so the default is computed once
                    default = create_prior_params(inputs=self.inputs, n_corr=self.n_corr[i], nugget_type=nugget_type)
                params = default
            self._mogp_gpu.create_priors_for_emulator(i, *params)
            apply_mean_priors(self._mogp_gpu.emulator(i), pr, self._analytic_mean)

    inputs = property(lambda self: self._mogp_gpu.inputs())
    targets = property(lambda self: np.array(self._mogp_gpu.targets()))
    D = property(lambda self: self._mogp_gpu.D())
    n = property(lambda self: self._mogp_gpu.n())
    nugget_type = property(lambda self: self._mogp_gpu.get_nugget_type())
    nugget = property(lambda self: self._mogp_gpu.get_nugget_size())
    n_params = property(lambda self: self._mogp_gpu.n_data_params())
    n_emulators = property(lambda self: self._mogp_gpu.n_emulators())
    n_corr = property(lambda self: self._mogp_gpu.n_corr_params())

    @property
    def emulators(self):
        return [GaussianProcessGPU.from_cpp(self._mogp_gpu.emulator(i)) for i in range(self.n_emulators)]

    def reset_fit_status(self):
        self._mogp_gpu.reset_fit_status()

    def get_indices_fit(self):
        return self._mogp_gpu.get_fitted_indices()

    def get_indices_not_fit(self):
        return self._mogp_gpu.get_unfitted_indices()

    def fit(self, thetas):
        self._mogp_gpu.fit(np.asarray(thetas, dtype=np.float64))

    def fit_record(self):
        """Per-emulator fit results in one place -- what one rank contributes to the single gather of a sharded fit
        (dist.ShardedMultiOutputGP; fitting.hpp:111-117): fit status, current log-posterior, nugget in use and
        theta = [mean parameters, correlation / covariance (/ nugget) parameters]; None / nan where not fit."""
        ok, logpost, nugget, theta = [], [], [], []
        for i in range(self.n_emulators):
            em = self._mogp_gpu.emulator(i)
            fitted = bool(em.theta_fit_status())
            ok.append(fitted)
            nugget.append(float(em.get_nugget_size()))
            if fitted:
                th = em.get_theta()
                vec = np.concatenate([th.get_mean(), th.get_data()])
                theta.append(vec)
                logpost.append(float(em.get_logpost(vec)))          # cached: theta is the current one
            else:
                theta.append(None)
                logpost.append(np.nan)
        return {"fit_ok": ok, "logpost": logpost, "nugget": nugget, "theta": theta}

    def fit_emulator(self, index, theta):
        self._mogp_gpu.fit_emulator(index, np.asarray(theta, dtype=np.float64))

    def _nuggets(self):
        return np.array([self._mogp_gpu.emulator(i).get_nugget_size() for i in range(self.n_emulators)])

    def predict(self, testing, unc=True, deriv=True, include_nugget=True, allow_not_fit=False, processes=None, full_cov=False):
        not_fit = self.get_indices_not_fit()
        if not allow_not_fit and len(not_fit) > 0:
            raise ValueError("Hyperparameters have not been fit for this Gaussian Process")
        testing = np.ascontiguousarray(np.array(testing, dtype=np.float64))
        if testing.ndim == 1:
            testing = testing.reshape(-1, 1) if self.D == 1 else testing.reshape(1, -1)
        assert testing.ndim == 2, "testing must be a 2D array"
        m, D = testing.shape
        assert D == self.D, "second dimension of testing must be the same as the number of input parameters"
        means = np.zeros((self.n_emulators, m))
        uncs = np.zeros((self.n_emulators, m))
        derivs = np.zeros((self.n_emulators, m, self.D))
        if unc and full_cov:
            # (n_emulators, m, m) covariances, MultiOutputGP.predict(full_cov=True) of the CPU class
            uncs = np.zeros((self.n_emulators, m, m))
            self._mogp_gpu.predict_full_cov(testing, means, uncs)
            if include_nugget:
                uncs[:, np.arange(m), np.arange(m)] += self._nuggets()[:, None]
        elif unc:
            self._mogp_gpu.predict_variance_batch(testing, means, uncs)
            if include_nugget:
                # per-emulator nugget actually in use (adaptive / fit values differ between emulators)
                uncs += self._nuggets()[:, None]
            np.maximum(uncs, 0., out=uncs)
        else:
            self._mogp_gpu.predict_batch(testing, means)
        if deriv:
            self._mogp_gpu.predict_deriv(testing, derivs)
        for idx in (not_fit if allow_not_fit else []):
            means[idx] = np.nan
            uncs[idx] = np.nan
            derivs[idx] = np.nan
        return PredictResult(mean=means, unc=uncs, deriv=derivs)

    def __call__(self, testing, processes=None):
        return self.predict(testing, unc=False, deriv=False, processes=processes)[0]

    def __str__(self):
        return ("Multi-Output Gaussian Process with:\n" + str(self.n_emulators) + " emulators\n" +
                str(self.n) + " training examples\n" + str(self.D) + " input variables")
