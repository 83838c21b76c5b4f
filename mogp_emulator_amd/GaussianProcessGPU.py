"""
``GaussianProcessGPU`` -- host-side mirror of mogp_emulator/GaussianProcessGPU.py:208-667 on top
of the gfx950 backend.  Same constructor arguments, properties and method names; the numerical
work (covariance build, factorisation, solves, gradient, predictions) all happens in
``libmogp_hip.so`` through ``LibGPGPU.DenseGP_GPU``.

Deliberate differences from the reference wrapper (all follow the CPU ``GaussianProcess`` oracle,
SURVEY.md section 7 "quirks that must NOT be copied"):
  * predictive variances are clipped at zero like GaussianProcess.py:918-920;
  * works on NumPy >= 2 (the reference's ``np.array(copy=False)`` raises there);
  * pickling works (state = inputs, targets, settings, theta -> refit on load);
  * mean functions: ``None``, a number, ``"c"`` or sums of ``c*x[i]^p`` terms are accepted directly
    (the symbolic MeanFunction algebra of the reference is out of scope).
"""
import re

import numpy as np

from . import LibGPGPU
from .Kernel import KernelBase, Matern52, ProductMat52, SquaredExponential, UniformMat52, UniformSqExp
from .Priors import GammaPrior, GPPriors, InvGammaPrior, LogNormalPrior, MeanPriors, PriorDist, WeakPrior


class GPUUnavailableError(RuntimeError):
    """Raised when the GPU, or the GPU library, is unavailable."""


class PredictResult(dict):
    """(mean, unc, deriv) container with dict, attribute and positional access, same behaviour as
    GaussianProcess.py:948-1026."""
    _order = ("mean", "unc", "deriv")

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    def __getitem__(self, key):
        if isinstance(key, bool) or not isinstance(key, (int, str)):
            raise KeyError(key)
        if isinstance(key, int):
            if not 0 <= key < 3:
                raise KeyError(key)
            key = self._order[key]
        return dict.__getitem__(self, key)

    def __iter__(self):
        return iter([dict.__getitem__(self, k) for k in self._order])

    def __repr__(self):
        if not self.keys():
            return self.__class__.__name__ + "()"
        width = max(len(k) for k in self._order) + 1
        return "\n".join(k.rjust(width) + ": " + repr(self[k]) for k in self._order)


def ndarray_coerce_type_and_flags(arr):
    """float64, C-contiguous, writeable ndarray holding the data of ``arr``."""
    out = np.ascontiguousarray(np.asarray(arr), dtype=np.float64)
    return out if out.flags["WRITEABLE"] else out.copy()


_TERM = re.compile(r"x\[(\d+)\]")
_POW = re.compile(r"\^(\d+)")


def parse_meanfunc_formula(formula):
    """Turn a canonical mean-function string into a native mean-function object:
    ``"c"`` -> ConstMeanFunc, a number -> FixedMeanFunc, ``c+c*x[0]+c*x[1]^2`` -> PolyMeanFunc.
    Returns None if nothing could be recognised; cross terms raise NotImplementedError
    (GaussianProcessGPU.py:57-110)."""
    text = str(formula).replace(" ", "")
    if text == "c":
        return LibGPGPU.ConstMeanFunc()
    try:
        return LibGPGPU.FixedMeanFunc(float(text))
    except ValueError:
        pass
    dims_powers = []
    for term in text.split("+"):
        idx = [int(i) for i in _TERM.findall(term)]
        if not idx:
            continue
        if len(set(idx)) != 1:
            raise NotImplementedError("Cross terms, e.g. x[0]*x[1] not implemented in GPU version.")
        power = len(idx) + sum(int(p) - 1 for p in _POW.findall(term))
        dims_powers.append([idx[0], power])
    return LibGPGPU.PolyMeanFunc(dims_powers) if dims_powers else None


def interpret_nugget(nugget):
    """(LibGPGPU.nugget_type, size) from ``"adaptive"`` / ``"fit"`` / ``"pivot"`` / non-negative number (``"pivot"`` is the
    CPU class's pivoted-Cholesky mode, GPParams.py:185-186, which the reference's GPU class does not have)."""
    if not isinstance(nugget, (str, float)):
        try:
            nugget = float(nugget)
        except TypeError:
            raise TypeError("nugget parameter must be a string or a non-negative float")
    if isinstance(nugget, str):
        if nugget not in ("adaptive", "fit", "pivot"):
            raise ValueError("nugget must be a string set to 'adaptive', 'fit', 'pivot', or a float")
        return getattr(LibGPGPU.nugget_type, nugget), 0.
    if nugget < 0.:
        raise ValueError("nugget parameter must be non-negative")
    return LibGPGPU.nugget_type.fixed, nugget


def _native_prior(prior):
    """(prior_type, [shape, scale]) of a ``Priors.*`` object or of one of the native prior objects ``LibGPGPU``
    re-exports (bindings.cu:458-545); only an actual weak prior -- never an unrecognised subclass -- maps to Weak."""
    for host_cls, native_cls, kind in ((InvGammaPrior, LibGPGPU.InvGammaPrior, LibGPGPU.prior_type.InvGamma),
                                       (GammaPrior, LibGPGPU.GammaPrior, LibGPGPU.prior_type.Gamma),
                                       (LogNormalPrior, LibGPGPU.LogNormalPrior, LibGPGPU.prior_type.LogNormal)):
        if isinstance(prior, (host_cls, native_cls)):
            return (kind, [float(prior.shape), float(prior.scale)])
    if prior is None or type(prior) in (WeakPrior, LibGPGPU.WeakPrior):
        return (LibGPGPU.prior_type.Weak, [0., 0.])
    raise TypeError("Unknown prior type {} for C++/GPU implementation".format(type(prior)))


def create_prior_params(**kwargs):
    """[n_corr, [(type,[shape,scale])...], (type,[..]) cov, (type,[..]) nugget] for
    ``create_gppriors``; built either from ``newpriors=`` (GPPriors or dict of its arguments) or from
    ``inputs=, n_corr=, nugget_type=`` (default priors)."""
    if all(k in kwargs for k in ("inputs", "n_corr", "nugget_type")):
        priors = GPPriors.default_priors(kwargs["inputs"], kwargs["n_corr"], kwargs["nugget_type"])
    elif "newpriors" in kwargs:
        priors = kwargs["newpriors"]
        if not isinstance(priors, GPPriors):
            try:
                priors = GPPriors(**priors)
            except TypeError:
                raise TypeError("Provided arguments for priors are not valid inputs for a GPPriors object.")
    else:
        raise TypeError("Unrecognized keyword arguments for create_prior_params "
                        " - should be 'newpriors' or ('inputs','n_corr','nugget_type')")
    return [priors.n_corr, [_native_prior(p) for p in priors.corr], _native_prior(priors.cov),
            _native_prior(priors.nugget)]


def apply_mean_priors(native, priors, analytic_mean):
    """hand the MeanPriors part of a GPPriors (or of its dict form) to a native emulator; weak priors reset it"""
    mp = None
    if isinstance(priors, GPPriors):
        mp = priors.mean
    elif isinstance(priors, dict) and priors.get("mean") is not None:
        mp = priors["mean"] if isinstance(priors["mean"], MeanPriors) else MeanPriors(*priors["mean"])
    if mp is None or mp.has_weak_priors:
        if analytic_mean:
            native.set_mean_priors(0, np.zeros(1), np.zeros(1), np.zeros(1), 0.)
        return
    if not analytic_mean:
        raise NotImplementedError("mean-function priors need the analytic mean function: construct the emulator with "
                                  "analytic_mean=True (the reference GPU path optimises the coefficients inside theta)")
    native.set_mean_priors(*mp.native_params())


def _resolve_kernel(kernel):
    """name or kernel object -> (native enum, kernel object).  SquaredExponential / Matern52 are the reference GPU
    kernels (GaussianProcessGPU.py:263-277); ProductMat52 / UniformSqExp / UniformMat52 are CPU-only there."""
    for cls in (SquaredExponential, Matern52, ProductMat52, UniformSqExp, UniformMat52):
        if kernel == cls.native_name or type(kernel) is cls:
            return getattr(LibGPGPU.kernel_type, cls.native_name), cls()
    raise ValueError("GPU implementation requires kernel to be one of SquaredExponential, Matern52, ProductMat52, "
                     "UniformSqExp or UniformMat52")


def _resolve_mean(mean):
    """None, a native mean function, a formula string, or an object whose ``str()`` is a canonical formula (the
    reference's MeanBase); anything else -- numbers included -- is a ValueError (GaussianProcessGPU.py:279-300)."""
    if mean is None:
        return LibGPGPU.ZeroMeanFunc()
    if isinstance(mean, LibGPGPU.BaseMeanFunc):
        return mean
    formula_like = isinstance(mean, str) or (hasattr(mean, "get_n_params") and not isinstance(mean, (int, float, complex)))
    if not formula_like:
        raise ValueError("provided mean function must be a formula string, a native mean function, or None")
    native = parse_meanfunc_formula(str(mean))
    if native is None:
        raise ValueError("GPU implementation was unable to parse mean function formula {}.".format(mean))
    return native


class GaussianProcessGPU(object):
    def __init__(self, inputs, targets, mean=None, kernel=SquaredExponential(), priors=None, nugget="adaptive",
                 inputdict={}, use_patsy=True, max_batch_size=2000, analytic_mean=False):
        if not LibGPGPU.HAVE_LIBGPGPU:
            raise RuntimeError("Cannot construct GaussianProcessGPU: The GPU library (libgpgpu) could not be loaded")
        if not LibGPGPU.gpu_usable():
            raise RuntimeError("Cannot construct GaussianProcessGPU: A compatible GPU could not be found")
        inputs = ndarray_coerce_type_and_flags(inputs)
        if inputs.ndim == 1:
            inputs = inputs.reshape(-1, 1)
        assert inputs.ndim == 2
        targets = ndarray_coerce_type_and_flags(targets)
        assert targets.ndim == 1
        assert targets.shape[0] == inputs.shape[0]
        self._inputs, self._targets = inputs, targets
        self._max_batch_size = int(max_batch_size)
        self.mean = _resolve_mean(mean)
        self.kernel_type, self.kernel = _resolve_kernel(kernel)
        self._nugget_type, self._init_nugget_size = interpret_nugget(nugget)
        self._priors_arg = priors
        # analytic_mean=True: mean coefficients integrated out with weak priors (GaussianProcess.py:640-700)
        self._analytic_mean = bool(analytic_mean)
        self._densegp_gpu = None
        self._init_gpu()
        self._set_priors(priors)

    @classmethod
    def from_cpp(cls, denseGP_GPU):
        obj = cls.__new__(cls)
        obj._densegp_gpu = denseGP_GPU
        obj._inputs, obj._targets = denseGP_GPU.inputs(), denseGP_GPU.targets()
        obj._nugget_type = denseGP_GPU.get_nugget_type()
        obj._init_nugget_size = denseGP_GPU.get_nugget_size()
        obj.kernel_type, obj.kernel = _resolve_kernel(str(denseGP_GPU.get_kernel_type()).split(".")[1])
        obj.mean = denseGP_GPU.get_meanfunc()
        obj._max_batch_size = 2000
        obj._priors_arg = None
        obj._analytic_mean = False
        return obj

    def _init_gpu(self):
        if self._densegp_gpu is None:
            self._densegp_gpu = LibGPGPU.DenseGP_GPU(self._inputs, self._targets, self._max_batch_size, self.mean,
                                                     self.kernel_type, self._nugget_type, self._init_nugget_size,
                                                     analytic_mean=getattr(self, "_analytic_mean", False))

    def _set_priors(self, newpriors=None):
        if newpriors:
            params = create_prior_params(newpriors=newpriors)
        else:
            params = create_prior_params(inputs=self.inputs, n_corr=self.n_corr, nugget_type=self.nugget_type)
        assert params[0] == self.n_corr, "bad number of correlation lengths in new GPPriors object"
        self._densegp_gpu.create_gppriors(*params)
        apply_mean_priors(self._densegp_gpu, newpriors, getattr(self, "_analytic_mean", False))

    # -- read-only views of the native state --------------------------------------------------------
    priors = property(lambda self: self._densegp_gpu.get_gppriors())
    inputs = property(lambda self: self._densegp_gpu.inputs())
    targets = property(lambda self: self._densegp_gpu.targets())
    n = property(lambda self: self._densegp_gpu.n())
    D = property(lambda self: self._densegp_gpu.D())
    n_corr = property(lambda self: self._densegp_gpu.n_corr())

    @property
    def n_params(self):
        th = self._densegp_gpu.get_theta()
        return th.get_n_data() + th.get_n_mean()

    @property
    def nugget_type(self):
        return str(self._nugget_type).split(".")[1]

    @property
    def nugget(self):
        if self.nugget_type == "pivot":
            return None                # GPParams.nugget, GPParams.py:444-445: no nugget with pivoting
        return self._densegp_gpu.get_nugget_size()

    @nugget.setter
    def nugget(self, nugget):
        self._nugget_type, size = interpret_nugget(nugget)
        self._densegp_gpu.set_nugget_type(self._nugget_type)
        self._densegp_gpu.set_nugget_size(size)

    @property
    def theta(self):
        return self._densegp_gpu.get_theta()

    @theta.setter
    def theta(self, theta):
        if theta is None:
            self._densegp_gpu.reset_theta_fit_status()
        else:
            self.fit(theta)

    @property
    def L(self):
        out = np.zeros((self.n, self.n))
        self._densegp_gpu.get_cholesky_lower(out)
        return np.tril(out.T)

    @property
    def P(self):
        """pivot order of the current fit (``Kinv.P`` of the CPU class with ``nugget="pivot"``): K[P][:, P] = L L^T"""
        return self._densegp_gpu.get_pivot()[0]

    @property
    def pivot_rank(self):
        """number of pivots the last pivoted factorisation accepted (n unless design points repeat)"""
        return self._densegp_gpu.get_pivot()[1]

    @property
    def Kinv_t(self):
        if not self._densegp_gpu.theta_fit_status():
            return None
        out = np.zeros(self.n)
        self._densegp_gpu.get_invQt(out)
        return out

    @property
    def current_logpost(self):
        if not self._densegp_gpu.theta_fit_status():
            return None
        th = self.theta
        return self.logposterior(np.concatenate([th.get_mean(), th.get_data()]))

    def get_K_matrix(self):
        out = np.zeros((self.n, self.n))
        self._densegp_gpu.get_K(out)
        return out

    # -- fit / objective ------------------------------------------------------------------------------
    def fit(self, theta):
        if isinstance(theta, LibGPGPU.GPParameters):
            self._densegp_gpu.fit(theta)
        else:
            self._densegp_gpu.fit(ndarray_coerce_type_and_flags(theta))

    def logposterior(self, theta):
        return self._densegp_gpu.get_logpost(ndarray_coerce_type_and_flags(theta))

    def logpost_deriv(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        assert theta.shape == (self.n_params,), "bad shape for new parameters"
        cur = self.theta
        stale = not cur.data_has_been_set() or not np.allclose(
            theta, np.concatenate([cur.get_mean(), cur.get_data()]), rtol=1.e-10, atol=1.e-15)
        if stale:
            self.fit(theta)
        out = np.zeros(self.n_params)
        self._densegp_gpu.logpost_deriv(out)
        return out

    def logpost_hessian(self, theta):
        raise GPUUnavailableError("The Hessian calculation is not currently implemented in the GPU version of MOGP.")

    # -- predict ------------------------------------------------------------------------------------------
    def predict(self, testing, unc=True, deriv=True, include_nugget=True, full_cov=False):
        if not self.theta.data_has_been_set():
            raise ValueError("hyperparameters have not been fit for this Gaussian Process")
        testing = ndarray_coerce_type_and_flags(testing)
        if testing.ndim == 1:
            testing = testing.reshape(-1, 1) if self.D == 1 else testing.reshape(1, -1)
        assert testing.ndim == 2
        m, D = testing.shape
        assert D == self.D
        if unc and full_cov:
            # CPU-class feature (GaussianProcess.py:899-911): (m, m) covariance, nugget on the diagonal, not clipped
            means, cov = np.zeros(m), np.zeros((m, m))
            self._densegp_gpu.predict_full_cov(testing, means, cov)
            if include_nugget and self.nugget_type != "pivot":      # GaussianProcess.py:904
                cov[np.diag_indices(m)] += self.nugget
            derivs = None
            if deriv:
                derivs = np.zeros((m, self.D))
                for lo in range(0, m, self._max_batch_size):
                    self._densegp_gpu.predict_deriv(testing[lo:lo + self._max_batch_size], derivs[lo:lo + self._max_batch_size])
            return PredictResult(mean=means, unc=cov, deriv=derivs)
        step = self._max_batch_size
        means = np.zeros(m)
        variances = np.zeros(m) if unc else None
        derivs = np.zeros((m, self.D)) if deriv else None
        for lo in range(0, m, step):
            chunk = testing[lo:lo + step]
            if unc:
                self._densegp_gpu.predict_variance_batch(chunk, means[lo:lo + step], variances[lo:lo + step])
            else:
                self._densegp_gpu.predict_batch(chunk, means[lo:lo + step])
            if deriv:
                self._densegp_gpu.predict_deriv(chunk, derivs[lo:lo + step])
        if unc:
            if include_nugget and self.nugget_type != "pivot":      # GaussianProcess.py:915
                variances += self.nugget
            np.maximum(variances, 0., out=variances)      # CPU oracle clips, GaussianProcess.py:918-920
        return PredictResult(mean=means, unc=variances, deriv=derivs)

    def __call__(self, testing):
        return self.predict(testing, unc=False, deriv=False)[0]

    def __str__(self):
        return "Gaussian Process with {} training examples and {} input variables".format(self.n, self.D)

    # -- pickling: drop the native handle, rebuild + refit on load ------------------------------------------
    def __getstate__(self):
        th = self.theta
        theta = np.concatenate([th.get_mean(), th.get_data()]) if th.data_has_been_set() else None
        # the CURRENT nugget setting (the ``nugget`` setter may have changed it since construction); the mean function
        # pickles as its constructor arguments (libgpgpu.*MeanFunc.__reduce__)
        nugget_size = self._densegp_gpu.get_nugget_size() if self.nugget_type == "fixed" else 0.
        return dict(inputs=self._inputs, targets=self._targets, max_batch_size=self._max_batch_size,
                    kernel=self.kernel, nugget_type=self.nugget_type, nugget_size=nugget_size,
                    priors=self._priors_arg, theta=theta, mean=self.mean, analytic_mean=self._analytic_mean)

    def __setstate__(self, state):
        nugget = state["nugget_size"] if state["nugget_type"] == "fixed" else state["nugget_type"]
        self.__init__(state["inputs"], state["targets"], mean=state.get("mean"), kernel=state["kernel"], priors=state["priors"],
                      nugget=nugget, max_batch_size=state["max_batch_size"], analytic_mean=state.get("analytic_mean", False))
        if state["theta"] is not None:
            self.fit(state["theta"])
