"""
``libgpgpu`` -- drop-in replacement for the reference's pybind11 extension of the
same name (mogp_gpu/src/bindings.cu:13-621), backed by the gfx950 C ABI
``libmogp_hip.so`` through ctypes (``_capi``).

``mogp_emulator/LibGPGPU.py:5-14`` does ``from libgpgpu import *``; putting this
module on ``sys.path`` under that name is the whole integration.
The exported names, argument orders, in-place output-buffer conventions and the
``RuntimeError`` behaviour follow the bindings cited next to each definition.

No CPU fallback exists here: import fails if the shared library is missing.
"""
import ctypes

import numpy as np

from . import _capi
from ._capi import check, dptr, iptr

_lib = _capi.load()

__all__ = [
    "have_compatible_device", "fit_GP_MAP", "kernel_type", "nugget_type", "prior_type",
    "BaseMeanFunc", "ZeroMeanFunc", "FixedMeanFunc", "ConstMeanFunc", "PolyMeanFunc",
    "GPParameters", "DenseGP_GPU", "MultiOutputGP_GPU", "GPPriors",
    "WeakPrior", "InvGammaPrior", "GammaPrior", "LogNormalPrior",
    "BaseTransform", "CovTransform", "CorrTransform",
    "SquaredExponentialKernel", "Matern52Kernel", "MeanPriors", "set_fit_options", "set_device", "device_count", "pivot_cholesky",
]


# --------------------------------------------------------------------------------------
# enums (bindings.cu:585-598; integer values from types.hpp:29-35)
# --------------------------------------------------------------------------------------
class _EnumMeta(type):
    def __iter__(cls):
        return iter(cls._members.values())

    def __call__(cls, value):
        if isinstance(value, cls):
            return value
        try:
            return cls._by_value[int(value)]
        except (KeyError, TypeError, ValueError):
            raise ValueError("%r is not a valid %s" % (value, cls.__name__))


def _make_enum(name, members):
    cls = _EnumMeta(name, (), {"_members": {}, "_by_value": {}})

    def _repr(self):
        return "%s.%s" % (name, self.name)

    cls.__str__ = _repr
    cls.__repr__ = lambda self: "<%s: %d>" % (_repr(self), self.value)
    cls.__int__ = lambda self: self.value
    cls.__index__ = lambda self: self.value
    cls.__eq__ = lambda self, other: isinstance(other, cls) and other.value == self.value
    cls.__ne__ = lambda self, other: not cls.__eq__(self, other)
    cls.__hash__ = lambda self: hash((name, self.value))
    for mname, mval in members:
        obj = object.__new__(cls)
        obj.name, obj.value = mname, mval
        cls._members[mname] = obj
        cls._by_value[mval] = obj
        setattr(cls, mname, obj)
    return cls


# 0, 1: the reference enum (types.hpp:29-35).  2-4: the kernels the reference only has on the CPU (Kernel.py:946-997,
# SURVEY 8f row 4); the uniform kernels share one correlation length over all inputs (n_corr = 1).
kernel_type = _make_enum("kernel_type", [("SquaredExponential", 0), ("Matern52", 1), ("ProductMat52", 2),
                                         ("UniformSqExp", 3), ("UniformMat52", 4)])
# 3: nugget="pivot" of the CPU class (GPParams.py:185-186), not in the reference's GPU enum (types.hpp:29-35)
nugget_type = _make_enum("nugget_type", [("adaptive", 0), ("fit", 1), ("fixed", 2), ("pivot", 3)])
prior_type = _make_enum("prior_type", [("InvGamma", 0), ("Gamma", 1), ("LogNormal", 2), ("Weak", 3)])


def have_compatible_device():
    """bindings.cu:600 / util.hpp:40-47"""
    return bool(_lib.mogp_have_compatible_device())


def device_count():
    return int(_lib.mogp_device_count())


def set_device(i):
    """Select the HIP device for handles created afterwards (one process per GPU)."""
    check(_lib.mogp_set_device(int(i)))


def set_fit_options(max_iter=0, ftol=0., gtol=0., seed=0):
    check(_lib.mogp_set_fit_options(int(max_iter), float(ftol), float(gtol), int(seed)))


def _f64(a, ndim=None, name="array"):
    out = np.ascontiguousarray(a, dtype=np.float64)
    if ndim is not None and out.ndim != ndim:
        raise TypeError("%s must be a %d-D float64 array" % (name, ndim))
    return out


def _outbuf(a, name):
    """Caller-allocated output buffer filled in place (Eigen::Ref semantics, types.hpp:16-18)."""
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"] or not a.flags["WRITEABLE"]:
        raise TypeError("%s must be a writeable C-contiguous float64 ndarray" % name)
    return a


# --------------------------------------------------------------------------------------
# mean functions (bindings.cu:365-413)
# --------------------------------------------------------------------------------------
class BaseMeanFunc(object):
    _h = None

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.mogp_meanfunc_destroy(self._h)
            self._h = None

    def get_n_params(self):
        return int(_lib.mogp_meanfunc_n_params(self._h))

    def _call(self, fn, xs, params, rows):
        xs = _f64(xs)
        if xs.ndim == 1:
            xs = xs.reshape(1, -1)
        p = _f64(np.atleast_1d(params)) if np.size(params) else np.zeros(0)
        m, D = xs.shape
        out = np.zeros(rows(m, D))
        check(fn(self._h, dptr(xs), m, D, dptr(p) if p.size else None, int(p.size), dptr(out)))
        return out

    def mean_f(self, xs, params):
        return self._call(_lib.mogp_meanfunc_mean_f, xs, params, lambda m, D: (m,))

    def mean_deriv(self, xs, params):
        np_ = self.get_n_params()
        if np_ == 0:
            self._call(_lib.mogp_meanfunc_mean_f, xs, params, lambda m, D: (m,))   # length check
            return np.zeros((np.atleast_2d(xs).shape[0], 1))                        # meanfunc.hpp:68-75
        return self._call(_lib.mogp_meanfunc_mean_deriv, xs, params, lambda m, D: (np_, m))

    def mean_inputderiv(self, xs, params):
        return self._call(_lib.mogp_meanfunc_mean_inputderiv, xs, params, lambda m, D: (D, m))


# the native handle never travels: a pickled mean function is rebuilt from its constructor arguments
class ZeroMeanFunc(BaseMeanFunc):
    def __init__(self):
        self._h = _lib.mogp_meanfunc_zero()

    def __reduce__(self):
        return (ZeroMeanFunc, ())


class FixedMeanFunc(BaseMeanFunc):
    def __init__(self, value):
        self._value = float(value)
        self._h = _lib.mogp_meanfunc_fixed(self._value)

    def __reduce__(self):
        return (FixedMeanFunc, (self._value,))


class ConstMeanFunc(BaseMeanFunc):
    def __init__(self):
        self._h = _lib.mogp_meanfunc_const()

    def __reduce__(self):
        return (ConstMeanFunc, ())


class PolyMeanFunc(BaseMeanFunc):
    def __init__(self, dims_powers):
        dp = np.asarray(dims_powers, dtype=np.int32).reshape(-1, 2)
        self._dims = np.ascontiguousarray(dp[:, 0])
        self._pows = np.ascontiguousarray(dp[:, 1])
        self._h = _lib.mogp_meanfunc_poly(iptr(self._dims), iptr(self._pows), int(dp.shape[0]))

    def __reduce__(self):
        return (PolyMeanFunc, (np.stack([self._dims, self._pows], axis=1).tolist(),))


# --------------------------------------------------------------------------------------
# transforms and stand-alone priors (bindings.cu:458-545): host scalar objects
# --------------------------------------------------------------------------------------
class BaseTransform(object):
    pass


class CovTransform(BaseTransform):
    """sigma^2 = exp(theta): gpparams.hpp:72-100"""
    def raw_to_scaled(self, r):
        return np.exp(r)

    def scaled_to_raw(self, s):
        return np.log(s)

    def dscaled_draw(self, s):
        return s

    def d2scaled_draw2(self, s):
        return s


class CorrTransform(BaseTransform):
    """l = exp(-theta/2).  d2scaled_draw2 follows the CPU oracle (+s/4, GPParams.py:69-80); the
    C++ reference has the sign wrong (gpparams.hpp:59-61)."""
    def raw_to_scaled(self, r):
        return np.exp(-0.5 * np.asarray(r))

    def scaled_to_raw(self, s):
        return -2. * np.log(s)

    def dscaled_draw(self, s):
        return -0.5 * np.asarray(s)

    def d2scaled_draw2(self, s):
        return 0.25 * np.asarray(s)


class WeakPrior(object):
    def logp(self, x):
        return 0.

    def dlogpdx(self, x):
        return 0.

    def d2logpdx2(self, x):
        return 0.

    def dlogpdtheta(self, x, transform):
        """chain rule to the raw parameter (gppriors.hpp WeakPrior::dlogpdtheta; Priors.py:617-634)"""
        return float(self.dlogpdx(x) * transform.dscaled_draw(x))

    def d2logpdtheta2(self, x, transform):
        """Priors.py:648-666"""
        return float(self.d2logpdx2(x) * transform.dscaled_draw(x) ** 2 + self.dlogpdx(x) * transform.d2scaled_draw2(x))

    def sample(self, transform=None):
        return float(5. * (np.random.rand() - 0.5))


class _ShapeScalePrior(WeakPrior):
    def __init__(self, shape, scale):
        if not (shape > 0. and scale > 0.):
            raise RuntimeError("shape and scale parameters must be positive")
        self.shape, self.scale = float(shape), float(scale)

    def sample(self, transform=None):
        x = self.sample_x()
        return float(transform.scaled_to_raw(x)) if transform is not None else x


class InvGammaPrior(_ShapeScalePrior):
    def logp(self, x):
        from math import lgamma, log
        return self.shape * log(self.scale) - lgamma(self.shape) - (self.shape + 1.) * log(x) - self.scale / x

    def dlogpdx(self, x):
        return -(self.shape + 1.) / x + self.scale / x ** 2

    def d2logpdx2(self, x):
        return (self.shape + 1.) / x ** 2 - 2. * self.scale / x ** 3

    def sample_x(self):
        return float(self.scale / np.random.gamma(self.shape))


class GammaPrior(_ShapeScalePrior):
    def logp(self, x):
        from math import lgamma, log
        return -self.shape * log(self.scale) - lgamma(self.shape) + (self.shape - 1.) * log(x) - x / self.scale

    def dlogpdx(self, x):
        return (self.shape - 1.) / x - 1. / self.scale

    def d2logpdx2(self, x):
        return -(self.shape - 1.) / x ** 2

    def sample_x(self):
        return float(np.random.gamma(self.shape, self.scale))


class LogNormalPrior(_ShapeScalePrior):
    def logp(self, x):
        from math import log, pi
        return -0.5 * (log(x / self.scale) / self.shape) ** 2 - 0.5 * log(2. * pi) - log(x) - log(self.shape)

    def dlogpdx(self, x):
        from math import log
        return -log(x / self.scale) / self.shape ** 2 / x - 1. / x

    def d2logpdx2(self, x):
        from math import log
        return (-1. / self.shape ** 2 + log(x / self.scale) / self.shape ** 2 + 1.) / x ** 2

    def sample_x(self):
        return float(np.random.lognormal(np.log(self.scale), self.shape))


# --------------------------------------------------------------------------------------
# GPParameters (bindings.cu:415-456 / gpparams.hpp:102-238): host object
# --------------------------------------------------------------------------------------
class GPParameters(object):
    def __init__(self, n_mean=0, n_corr=1, nugget=nugget_type.fit, nugget_size=0.):
        self._n_mean, self._n_corr = int(n_mean), int(n_corr)
        self._nug_type = nugget_type(nugget)
        self._nug_size = float(nugget_size)
        self._has_data = False
        self._mean = np.zeros(self._n_mean)
        self._data = np.zeros(self.get_n_data())

    def get_n_data(self):
        return self._n_corr + 1 + int(self._nug_type == nugget_type.fit)

    def get_n_mean(self):
        return self._n_mean

    def get_n_corr(self):
        return self._n_corr

    def get_data(self):
        return self._data.copy()

    def set_data(self, new):
        new = np.array(new, dtype=np.float64).reshape(-1)
        if new.size != self.get_n_data():
            raise RuntimeError("New data not correct shape")
        self._data = new
        self._has_data = True
        self._mean = np.zeros(self._n_mean)
        if self._nug_type == nugget_type.adaptive:
            self._nug_size = 0.

    def get_mean(self):
        return self._mean.copy()

    def set_mean(self, new):
        self._mean = np.array(new, dtype=np.float64).reshape(-1)
        self._n_mean = self._mean.size

    def get_corr_raw(self):
        return self._data[:self._n_corr].copy()

    def get_corr(self):
        return np.exp(-0.5 * self._data[:self._n_corr])

    def set_corr(self, new):
        self._data[:self._n_corr] = -2. * np.log(np.asarray(new, dtype=np.float64))

    def _cov_index(self):
        return self.get_n_data() - (2 if self._nug_type == nugget_type.fit else 1)

    def get_cov(self):
        return float(np.exp(self._data[self._cov_index()])) if self._has_data else 0.

    def set_cov(self, cov):
        if not self._has_data:
            raise RuntimeError("Need to set data before setting covariance parameter")
        self._data[self._cov_index()] = np.log(cov)

    def get_nugget_type(self):
        return self._nug_type

    def set_nugget_type(self, t):
        self._nug_type = nugget_type(t)

    def get_nugget_size(self):
        if self._nug_type != nugget_type.fit:
            return self._nug_size
        return float(np.exp(self._data[-1]))

    def set_nugget_size(self, size):
        self._nug_size = float(size)
        if self._nug_type == nugget_type.fit and self._data.size:
            self._data[-1] = self._nug_size

    def unset_data(self):
        self._mean = np.zeros(self._n_mean)
        self._data = np.zeros(self.get_n_data())
        self._has_data = False

    def data_has_been_set(self):
        return self._has_data

    def test_same_shape(self, other):
        if isinstance(other, GPParameters):
            return (self._n_mean == other._n_mean and self._n_corr == other._n_corr
                    and self._nug_type == other._nug_type)
        return np.size(other) == self._n_mean + self.get_n_data()


# --------------------------------------------------------------------------------------
# GPPriors proxy (bindings.cu:547-566): the priors live inside the native emulator
# --------------------------------------------------------------------------------------
class GPPriors(object):
    """bindings.cu:528-556 / gppriors.hpp:318-510.  Two forms:

    * ``GPPriors(n_corr, nugget_type)`` -- the constructible container of the native module: ``set_corr`` / ``get_corr`` /
      ``set_cov`` / ``get_cov`` / ``set_nugget`` / ``create_corr_priors`` / ``create_cov_prior`` / ``make_prior`` /
      ``get_logp`` / ``get_dlogpdtheta`` / ``get_d2logpdtheta2`` / ``sample`` on host prior objects (O(D) scalar work;
      arithmetic of the CPU class, Priors.py:291-418);
    * ``GPPriors(emulator)`` (what ``DenseGP_GPU.get_gppriors()`` returns) -- a view on the priors that live inside a
      native emulator; ``get_logp`` / ``get_dlogpdtheta`` / ``sample`` go through the C ABI.
    ``DenseGP_GPU.set_gppriors(priors)`` accepts either."""

    def __init__(self, n_corr_or_owner, nug_type=None):
        if hasattr(n_corr_or_owner, "_h"):
            self._owner = n_corr_or_owner
            return
        self._owner = None
        self._n_corr = int(n_corr_or_owner)
        self._nug_type = nugget_type(nugget_type.fit if nug_type is None else nug_type)
        self._corr = []
        self._cov = None
        self._nug = None
        self._mean = None

    # -- view on a native emulator ---------------------------------------------------------------------------------
    def _data(self, theta):
        return theta.get_data() if isinstance(theta, GPParameters) else _f64(theta)

    # -- container ------------------------------------------------------------------------------------------------------
    def get_nugget_type(self):
        return self._nug_type if self._owner is None else self._owner.get_nugget_type()

    def set_corr(self, newcorr=None):
        """set_corr() appends n_corr weak priors (gppriors.hpp:356-360); set_corr(list) replaces them (:351-354)"""
        self._own_only()
        if newcorr is None:
            self._corr.extend(WeakPrior() for _ in range(self._n_corr))
        else:
            self._corr = list(newcorr)
            self._n_corr = len(self._corr)

    def get_corr(self):
        self._own_only()
        return list(self._corr)

    def set_cov(self, newcov=None):
        self._own_only()
        self._cov = WeakPrior() if newcov is None else newcov

    def get_cov(self):
        self._own_only()
        return self._cov

    def set_nugget(self, new=None):
        """only kept for a fitted nugget (gppriors.hpp:375-391); a (prior_type, [shape, scale]) pair is instantiated"""
        self._own_only()
        if self._nug_type == nugget_type.fit:
            if new is None:
                new = WeakPrior()
            elif isinstance(new, (tuple, list)):
                new = self.make_prior(*new)
            self._nug = new

    def get_nugget(self):
        self._own_only()
        return self._nug

    def get_mean(self):
        self._own_only()
        return self._mean

    def set_mean(self, mean=None):
        self._own_only()
        self._mean = MeanPriors() if mean is None else mean

    @staticmethod
    def make_prior(ptype, priorparams=()):
        """gppriors.hpp:473-488: anything that is not a two-parameter InvGamma / Gamma / LogNormal becomes a weak prior"""
        ptype = prior_type(ptype)
        params = list(priorparams)
        cls = {prior_type.InvGamma: InvGammaPrior, prior_type.Gamma: GammaPrior, prior_type.LogNormal: LogNormalPrior}.get(ptype)
        if cls is not None and len(params) == 2:
            return cls(params[0], params[1])
        return WeakPrior()

    def create_corr_priors(self, all_params):
        """appends one prior per (prior_type, [shape, scale]) pair (gppriors.hpp:490-497)"""
        self._own_only()
        for ptype, params in all_params:
            self._corr.append(self.make_prior(ptype, params))

    def create_cov_prior(self, params):
        self._own_only()
        self._cov = self.make_prior(*params)

    def _own_only(self):
        if self._owner is not None:
            raise RuntimeError("this GPPriors object is a view on an emulator's priors; build a GPPriors(n_corr, nugget_type) to edit")

    def _check_theta(self, theta):
        if not isinstance(theta, GPParameters):
            raise TypeError("theta must be a GPParameters object")
        if not theta.data_has_been_set() or theta.get_n_corr() != len(self._corr) or theta.get_nugget_type() != self._nug_type:
            raise RuntimeError("theta does not match the priors (n_corr, nugget type) or holds no data")
        if self._cov is None or (self._nug_type == nugget_type.fit and self._nug is None):
            raise RuntimeError("covariance / nugget priors have not been set")

    def _terms(self, theta):
        """(prior, scaled value, transform) of every data parameter in theta order"""
        self._check_theta(theta)
        corr = np.atleast_1d(theta.get_corr())
        out = [(pr, float(corr[i]), CorrTransform()) for i, pr in enumerate(self._corr)]
        out.append((self._cov, float(theta.get_cov()), CovTransform()))
        if self._nug_type == nugget_type.fit:
            out.append((self._nug, float(theta.get_nugget_size()), CovTransform()))
        return out

    def get_logp(self, theta):
        if self._owner is not None:
            data = self._data(theta)
            out = np.zeros(1)
            check(_lib.mogp_densegp_priors_logp(self._owner._h, dptr(data), int(data.size), dptr(out)))
            return float(out[0])
        return float(sum(pr.logp(x) for pr, x, _ in self._terms(theta)))

    def get_dlogpdtheta(self, theta):
        if self._owner is not None:
            data = self._data(theta)
            out = np.zeros(data.size)
            check(_lib.mogp_densegp_priors_dlogpdtheta(self._owner._h, dptr(data), int(data.size), dptr(out)))
            return out
        return np.array([pr.dlogpdtheta(x, tr) for pr, x, tr in self._terms(theta)])

    def get_d2logpdtheta2(self, theta):
        """diagonal of the Hessian with respect to the raw parameters (gppriors.hpp:439-456; Priors.py:356-391)"""
        self._own_only()
        return np.array([pr.d2logpdtheta2(x, tr) for pr, x, tr in self._terms(theta)])

    def sample(self):
        if self._owner is not None:
            out = np.zeros(self._owner.n_params() + _lib.mogp_densegp_n_mean(self._owner._h))
            check(_lib.mogp_densegp_priors_sample(self._owner._h, dptr(out)))
            return list(out)
        vals = list(self._mean.sample(CorrTransform())) if self._mean is not None else []      # gppriors.hpp:458-471
        vals += [pr.sample(CorrTransform()) for pr in self._corr]
        vals.append(self._cov.sample(CovTransform()))
        if self._nug_type == nugget_type.fit:
            vals.append(self._nug.sample(CovTransform()))
        return vals

    def native_params(self):
        """(n_corr, corr_params, cov_params, nug_params) for DenseGP_GPU.create_gppriors"""
        self._own_only()

        def spec(pr):
            for t, cls in ((prior_type.InvGamma, InvGammaPrior), (prior_type.Gamma, GammaPrior), (prior_type.LogNormal, LogNormalPrior)):
                if isinstance(pr, cls):
                    return (t, [pr.shape, pr.scale])
            return (prior_type.Weak, [0., 0.])
        return len(self._corr), [spec(pr) for pr in self._corr], spec(self._cov), spec(self._nug)


def _prior_spec(spec):
    t, p = spec
    p = list(p) + [0., 0.]
    return int(prior_type(t)), float(p[0]), float(p[1])


def _pack_priors(n_corr, corr_params, cov_params, nug_params):
    ct = np.zeros(max(n_corr, 1), dtype=np.int32)
    cp = np.zeros(2 * max(n_corr, 1))
    if len(corr_params) != n_corr:
        raise RuntimeError("number of correlation priors must equal n_corr")
    for d, spec in enumerate(corr_params):
        ct[d], cp[2 * d], cp[2 * d + 1] = _prior_spec(spec)
    cvt, a, b = _prior_spec(cov_params)
    cov = np.array([a, b])
    ngt, a, b = _prior_spec(nug_params)
    nug = np.array([a, b])
    return ct, cp, cvt, cov, ngt, nug


# --------------------------------------------------------------------------------------
# DenseGP_GPU (bindings.cu:14-257)
# --------------------------------------------------------------------------------------
class DenseGP_GPU(object):
    def __init__(self, inputs, targets, testing_size, meanfunc=None, kern=kernel_type.SquaredExponential,
                 nugtype=nugget_type.adaptive, nugsize=0., _borrowed=None, _parent=None, analytic_mean=False):
        if _borrowed is not None:
            self._h, self._owns, self._parent = _borrowed, False, _parent
            self._meanfunc = meanfunc
            return
        X = _f64(inputs, 2, "inputs")
        t = _f64(targets, 1, "targets")
        if t.shape[0] != X.shape[0]:
            raise RuntimeError("inputs and targets must have the same first dimension")
        if meanfunc is None:
            meanfunc = ZeroMeanFunc()
        self._meanfunc = meanfunc
        create = _lib.mogp_densegp_create_analytic_mean if analytic_mean else _lib.mogp_densegp_create
        self._h = create(dptr(X), X.shape[0], X.shape[1], dptr(t), int(testing_size), meanfunc._h,
                         int(kernel_type(kern)), int(nugget_type(nugtype)), float(nugsize))
        if not self._h:
            raise RuntimeError(_capi.last_error())
        self._owns, self._parent = True, None

    def __del__(self):
        if getattr(self, "_owns", False) and getattr(self, "_h", None):
            _lib.mogp_densegp_destroy(self._h)
            self._h = None

    # -- shape / data -------------------------------------------------------------------
    def n(self):
        return int(_lib.mogp_densegp_n(self._h))

    def D(self):
        return int(_lib.mogp_densegp_D(self._h))

    def n_corr(self):
        return int(_lib.mogp_densegp_n_corr(self._h))

    def n_params(self):
        return int(_lib.mogp_densegp_n_params(self._h))

    def inputs(self):
        out = np.zeros((self.n(), self.D()))
        check(_lib.mogp_densegp_inputs(self._h, dptr(out)))
        return out

    def targets(self):
        out = np.zeros(self.n())
        check(_lib.mogp_densegp_targets(self._h, dptr(out)))
        return out

    def set_mean_priors(self, q, b, Binv, Binv_b, logdetB):
        """MeanPriors of the analytic mean: b (q), B^-1 (q, q), B^-1 b (q), log|B|; q = 0 -> weak"""
        b, Binv, Binv_b = _f64(b, 1, "b"), np.ascontiguousarray(Binv, dtype=np.float64), _f64(Binv_b, 1, "Binv_b")
        check(_lib.mogp_densegp_set_mean_priors(self._h, int(q), dptr(b), dptr(Binv), dptr(Binv_b), float(logdetB)))

    def get_beta(self):
        """analytically fitted mean coefficients (analytic_mean=True only; theta.mean of the CPU class)"""
        nb = int(_lib.mogp_densegp_n_beta(self._h))
        out = np.zeros(max(nb, 1))
        check(_lib.mogp_densegp_get_beta(self._h, dptr(out)))
        return out[:nb].copy()

    # -- parameters -----------------------------------------------------------------------
    def theta_fit_status(self):
        return bool(_lib.mogp_densegp_theta_fit_status(self._h))

    def reset_theta_fit_status(self):
        check(_lib.mogp_densegp_reset_theta_fit_status(self._h))

    def get_theta(self):
        nm, nd = _lib.mogp_densegp_n_mean(self._h), _lib.mogp_densegp_n_data(self._h)
        p = GPParameters(nm, self.n_corr(), self.get_nugget_type(), 0.)
        data, mean = np.zeros(nd), np.zeros(max(nm, 1))
        check(_lib.mogp_densegp_get_theta(self._h, dptr(data), dptr(mean)))
        p._data, p._mean = data, mean[:nm].copy()
        p._has_data = self.theta_fit_status()
        p._nug_size = self.get_nugget_size()
        return p

    def get_gppriors(self):
        return GPPriors(self)

    def set_gppriors(self, priors):
        """bindings.cu:62-65: attach a GPPriors(n_corr, nugget_type) container (its distributions are copied into the
        native emulator; mean priors go through set_mean_priors)"""
        if not isinstance(priors, GPPriors) or priors._owner is not None:
            raise RuntimeError("set_gppriors: expected a GPPriors(n_corr, nugget_type) container")
        n_corr, corr, cov, nug = priors.native_params()
        if priors.get_cov() is None:
            raise RuntimeError("set_gppriors: the covariance prior has not been set")
        self.create_gppriors(n_corr, corr, cov, nug)

    def create_gppriors(self, n_corr, corr_params, cov_params, nug_params):
        ct, cp, cvt, cov, ngt, nug = _pack_priors(int(n_corr), corr_params, cov_params, nug_params)
        check(_lib.mogp_densegp_create_gppriors(self._h, int(n_corr), iptr(ct), dptr(cp), cvt, dptr(cov), ngt, dptr(nug)))

    def get_nugget_size(self):
        return float(_lib.mogp_densegp_get_nugget_size(self._h))

    def set_nugget_size(self, v):
        check(_lib.mogp_densegp_set_nugget_size(self._h, float(v)))

    def get_nugget_type(self):
        return nugget_type(_lib.mogp_densegp_get_nugget_type(self._h))

    def set_nugget_type(self, t):
        check(_lib.mogp_densegp_set_nugget_type(self._h, int(nugget_type(t))))

    def get_kernel_type(self):
        return kernel_type(_lib.mogp_densegp_get_kernel_type(self._h))

    def get_kernel(self):
        return _KERNEL_OBJECTS[int(self.get_kernel_type())]()

    def get_meanfunc(self):
        return self._meanfunc

    # -- fit / objective --------------------------------------------------------------------
    @staticmethod
    def _theta_vec(theta):
        if isinstance(theta, GPParameters):
            return np.concatenate([theta.get_mean(), theta.get_data()])
        return _f64(np.atleast_1d(theta), 1, "theta")

    def fit(self, theta):
        th = self._theta_vec(theta)
        check(_lib.mogp_densegp_fit(self._h, dptr(th), int(th.size)))

    def get_logpost(self, theta):
        th = self._theta_vec(theta)
        out = np.zeros(1)
        check(_lib.mogp_densegp_get_logpost(self._h, dptr(th), int(th.size), dptr(out)))
        return float(out[0])

    def logpost_deriv(self, result):
        _outbuf(result, "result")
        check(_lib.mogp_densegp_logpost_deriv(self._h, dptr(result), int(result.size)))

    # -- predict ------------------------------------------------------------------------------
    def _testing(self, testing):
        x = _f64(testing)
        if x.ndim == 1:
            x = x.reshape(1, -1)
        if x.ndim != 2:
            raise TypeError("testing must be a 2-D array")
        return x

    def predict(self, testing):
        x = _f64(testing).reshape(-1)
        out = np.zeros(1)
        check(_lib.mogp_densegp_predict(self._h, dptr(x), int(x.size), dptr(out)))
        return float(out[0])

    def predict_variance(self, testing, var):
        x = _f64(testing).reshape(-1)
        _outbuf(var, "var")
        mean, v = np.zeros(1), np.zeros(1)
        check(_lib.mogp_densegp_predict_variance(self._h, dptr(x), int(x.size), dptr(mean), dptr(v)))
        var.reshape(-1)[0] = v[0]
        return float(mean[0])

    def predict_batch(self, testing, result):
        x = self._testing(testing)
        _outbuf(result, "result")
        check(_lib.mogp_densegp_predict_batch(self._h, dptr(x), x.shape[0], x.shape[1], dptr(result), int(result.size)))

    def predict_variance_batch(self, testing, mean, var):
        x = self._testing(testing)
        _outbuf(mean, "mean")
        _outbuf(var, "var")
        check(_lib.mogp_densegp_predict_variance_batch(self._h, dptr(x), x.shape[0], x.shape[1], dptr(mean), dptr(var),
                                                       int(min(mean.size, var.size))))

    def predict_deriv(self, testing, result):
        x = self._testing(testing)
        _outbuf(result, "result")
        if result.ndim != 2:
            raise RuntimeError("predict_deriv: the result buffer passed was the wrong shape to hold the result")
        check(_lib.mogp_densegp_predict_deriv(self._h, dptr(x), x.shape[0], x.shape[1], dptr(result), result.shape[0],
                                              result.shape[1]))

    def predict_full_cov(self, testing, mean, cov):
        """mean (m,), cov (m, m): full predictive covariance WITHOUT nugget (GaussianProcess.py:899-911)"""
        x = self._testing(testing)
        _outbuf(mean, "mean")
        _outbuf(cov, "cov")
        if mean.size < x.shape[0] or cov.size < x.shape[0] ** 2:
            raise RuntimeError("predict_full_cov: The result buffer passed was too small to hold the covariance")
        check(_lib.mogp_densegp_predict_full_cov(self._h, dptr(x), x.shape[0], x.shape[1], dptr(mean), dptr(cov)))

    def implausibility(self, testing, obs, obs_var=0., discrepancy=0., include_nugget=True):
        """|obs - E f(x)| / sqrt(Var f(x) [+ nugget] + discrepancy + obs_var) for every row of testing, computed on
        the device behind the prediction (HistoryMatching.py:262-276)."""
        x = self._testing(testing)
        out = np.zeros(x.shape[0])
        check(_lib.mogp_densegp_implausibility(self._h, dptr(x), x.shape[0], x.shape[1], float(obs), float(obs_var),
                                               float(discrepancy), int(bool(include_nugget)), dptr(out)))
        return out

    def loo_variance(self):
        """leave-one-out predictive variance 1/[K^-1]_ii at every training input (MICEFastGP.fast_predict for all indices)"""
        out = np.zeros(self.n())
        check(_lib.mogp_densegp_loo_variance(self._h, dptr(out)))
        return out

    # -- matrices -------------------------------------------------------------------------------
    def _fill_nn(self, fn, out, name):
        _outbuf(out, name)
        if out.size < self.n() * self.n():
            raise RuntimeError("%s: the buffer passed is too small" % name)
        check(fn(self._h, dptr(out)))

    def get_K(self, K_h):
        self._fill_nn(_lib.mogp_densegp_get_K, K_h, "get_K")

    def get_invQ(self, invQ_h):
        self._fill_nn(_lib.mogp_densegp_get_invQ, invQ_h, "get_invQ")

    def get_cholesky_lower(self, result):
        self._fill_nn(_lib.mogp_densegp_get_cholesky_lower, result, "get_cholesky_lower")

    def get_pivot(self):
        """(P, rank) of the current fit: K[P][:, P] = L L^T (ChoInvPivot.P, linalg/cholesky.py:82-104); the identity and
        n unless the nugget type is ``pivot``."""
        P = np.zeros(self.n(), dtype=np.int32)
        rank = ctypes.c_int(0)
        check(_lib.mogp_densegp_get_pivot(self._h, iptr(P), ctypes.byref(rank)))
        return P, rank.value

    def get_invQt(self, invQt_h):
        _outbuf(invQt_h, "invQt_h")
        if invQt_h.size < self.n():
            raise RuntimeError("get_invQt: the buffer passed is too small")
        check(_lib.mogp_densegp_get_invQt(self._h, dptr(invQt_h)))


# --------------------------------------------------------------------------------------
# MultiOutputGP_GPU (bindings.cu:260-337)
# --------------------------------------------------------------------------------------
class MultiOutputGP_GPU(object):
    def __init__(self, inputs, targets, testing_size, meanfunc=None, kern=kernel_type.SquaredExponential,
                 nugtype=nugget_type.adaptive, nugsize=0., analytic_mean=False):
        X = _f64(inputs, 2, "inputs")
        T = np.ascontiguousarray(np.array([np.asarray(t, dtype=np.float64) for t in targets]))
        if T.ndim != 2 or T.shape[1] != X.shape[0]:
            raise RuntimeError("targets must have shape (n_emulators, n)")
        if meanfunc is None:
            meanfunc = ZeroMeanFunc()
        self._meanfunc = meanfunc
        create = _lib.mogp_mogp_create_analytic_mean if analytic_mean else _lib.mogp_mogp_create
        self._h = create(dptr(X), X.shape[0], X.shape[1], dptr(T), T.shape[0], int(testing_size), meanfunc._h,
                         int(kernel_type(kern)), int(nugget_type(nugtype)), float(nugsize))
        if not self._h:
            raise RuntimeError(_capi.last_error())

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.mogp_mogp_destroy(self._h)
            self._h = None

    def n(self):
        return int(_lib.mogp_mogp_n(self._h))

    def D(self):
        return int(_lib.mogp_mogp_D(self._h))

    def n_emulators(self):
        return int(_lib.mogp_mogp_n_emulators(self._h))

    def inputs(self):
        out = np.zeros((self.n(), self.D()))
        check(_lib.mogp_mogp_inputs(self._h, dptr(out)))
        return out

    def targets(self):
        out = np.zeros((self.n_emulators(), self.n()))
        check(_lib.mogp_mogp_targets(self._h, dptr(out)))
        return [row.copy() for row in out]

    def targets_at_index(self, index):
        return self.targets()[index]

    def emulator(self, index):
        h = _lib.mogp_mogp_emulator(self._h, int(index))
        if not h:
            raise RuntimeError(_capi.last_error())
        return DenseGP_GPU(None, None, 0, meanfunc=self._meanfunc, _borrowed=h, _parent=self)

    def n_data_params(self):
        return [self.emulator(i).n_params() for i in range(self.n_emulators())]

    def n_corr_params(self):
        return [self.emulator(0).n_corr()] * self.n_emulators()

    def get_nugget_type(self):
        return nugget_type(_lib.mogp_mogp_get_nugget_type(self._h))

    def get_nugget_size(self):
        return float(_lib.mogp_mogp_get_nugget_size(self._h))

    def _indices(self, fn):
        out = np.zeros(self.n_emulators(), dtype=np.int32)
        c = fn(self._h, iptr(out))
        return [int(i) for i in out[:c]]

    def get_fitted_indices(self):
        return self._indices(_lib.mogp_mogp_get_fitted_indices)

    def get_unfitted_indices(self):
        return self._indices(_lib.mogp_mogp_get_unfitted_indices)

    def reset_fit_status(self):
        check(_lib.mogp_mogp_reset_fit_status(self._h))

    def create_priors_for_emulator(self, emulator_index, n_corr, corr_params, cov_params, nug_params):
        ct, cp, cvt, cov, ngt, nug = _pack_priors(int(n_corr), corr_params, cov_params, nug_params)
        check(_lib.mogp_mogp_create_priors_for_emulator(self._h, int(emulator_index), int(n_corr), iptr(ct), dptr(cp), cvt,
                                                        dptr(cov), ngt, dptr(nug)))

    def fit_emulator(self, index, theta):
        th = DenseGP_GPU._theta_vec(theta)
        check(_lib.mogp_mogp_fit_emulator(self._h, int(index), dptr(th), int(th.size)))

    def fit(self, thetas):
        if len(thetas) and isinstance(thetas[0], GPParameters):
            thetas = [DenseGP_GPU._theta_vec(t) for t in thetas]
        th = _f64(thetas, 2, "thetas")
        check(_lib.mogp_mogp_fit(self._h, dptr(th), th.shape[0], th.shape[1]))

    def eval(self, thetas, grad=True):
        """Batched objective (+ gradient) of every emulator at its own theta: ONE device pass.
        Returns (logpost (n_out,), grad (n_out, n_params) or None, ok (n_out,) bool)."""
        th = _f64(thetas, 2, "thetas")
        f = np.zeros(th.shape[0])
        g = np.zeros(th.shape) if grad else None
        ok = np.zeros(th.shape[0], dtype=np.int32)
        check(_lib.mogp_mogp_eval(self._h, dptr(th), th.shape[0], th.shape[1], dptr(f), dptr(g), iptr(ok)))
        return f, g, ok.astype(bool)

    def _testing(self, testing):
        x = _f64(testing)
        if x.ndim == 1:
            x = x.reshape(1, -1)
        return x

    def predict(self, testing):
        x = self._testing(testing)[:1]
        out = np.zeros((self.n_emulators(), 1))
        check(_lib.mogp_mogp_predict_batch(self._h, dptr(x), 1, x.shape[1], dptr(out)))
        return out[:, 0]

    def predict_batch(self, testing, results):
        x = self._testing(testing)
        _outbuf(results, "results")
        if results.size < self.n_emulators() * x.shape[0]:
            raise RuntimeError("predict_batch: the result buffer passed was too small to hold the result")
        check(_lib.mogp_mogp_predict_batch(self._h, dptr(x), x.shape[0], x.shape[1], dptr(results)))

    def predict_variance_batch(self, testing, means, vars):
        x = self._testing(testing)
        _outbuf(means, "means")
        _outbuf(vars, "vars")
        if min(means.size, vars.size) < self.n_emulators() * x.shape[0]:
            raise RuntimeError("predict_variance_batch: The result buffer passed was too small to hold the variance")
        check(_lib.mogp_mogp_predict_variance_batch(self._h, dptr(x), x.shape[0], x.shape[1], dptr(means), dptr(vars)))

    def predict_deriv(self, testing, results):
        x = self._testing(testing)
        if isinstance(results, (list, tuple)):
            buf = np.zeros((self.n_emulators(), x.shape[0], x.shape[1]))
            check(_lib.mogp_mogp_predict_deriv(self._h, dptr(x), x.shape[0], x.shape[1], dptr(buf)))
            for r, b in zip(results, buf):
                r[...] = b
            return
        _outbuf(results, "results")
        if results.size < self.n_emulators() * x.size:
            raise RuntimeError("predict_deriv: the result buffer passed was the wrong shape to hold the result")
        check(_lib.mogp_mogp_predict_deriv(self._h, dptr(x), x.shape[0], x.shape[1], dptr(results)))

    def predict_full_cov(self, testing, means, covs):
        """means (n_out, m), covs (n_out, m, m) for every fitted emulator, one batched device pass (no nugget)"""
        x = self._testing(testing)
        _outbuf(means, "means")
        _outbuf(covs, "covs")
        if means.size < self.n_emulators() * x.shape[0] or covs.size < self.n_emulators() * x.shape[0] ** 2:
            raise RuntimeError("predict_full_cov: The result buffer passed was too small to hold the covariance")
        check(_lib.mogp_mogp_predict_full_cov(self._h, dptr(x), x.shape[0], x.shape[1], dptr(means), dptr(covs)))

    def implausibility(self, testing, obs, obs_var, discrepancy, include_nugget=True, rank=1):
        """history-matching score of every row of testing over all outputs, one number per point (device-fused)"""
        x = self._testing(testing)
        ne = self.n_emulators()
        z = np.ascontiguousarray(np.broadcast_to(np.asarray(obs, dtype=np.float64), (ne,)))
        ov = np.ascontiguousarray(np.broadcast_to(np.asarray(obs_var, dtype=np.float64), (ne,)))
        dc = np.ascontiguousarray(np.broadcast_to(np.asarray(discrepancy, dtype=np.float64), (ne,)))
        out = np.zeros(x.shape[0])
        check(_lib.mogp_mogp_implausibility(self._h, dptr(x), x.shape[0], x.shape[1], dptr(z), dptr(ov), dptr(dc),
                                            int(bool(include_nugget)), int(rank), dptr(out)))
        return out

    def predict_variance_batch_dev(self, d_testing, m, d_means, d_vars):
        """Device-pointer variant: inputs already resident in HBM, results stay in HBM."""
        check(_lib.mogp_mogp_predict_variance_batch_dev(self._h, int(d_testing), int(m), self.D(), int(d_means), int(d_vars)))

    def predict_dev(self, d_testing, m, d_means, d_vars=None, d_derivs=None):
        """Device pointers throughout: means (+ variances, + input derivatives (n_emulators, m, D)) of every emulator in one pass,
        results stay in HBM; any mean function; rows of emulators that are not fit are NaN."""
        check(_lib.mogp_mogp_predict_dev(self._h, int(d_testing), int(m), self.D(), int(d_means),
                                         int(d_vars) if d_vars else None, int(d_derivs) if d_derivs else None))


def fit_GP_MAP(gp, n_tries=15, theta0=()):
    """bindings.cu:602-605 / fitting.hpp:61-128"""
    th = _f64(np.atleast_1d(np.asarray(theta0, dtype=np.float64))) if np.size(theta0) else np.zeros(0)
    ptr = dptr(th) if th.size else None
    if isinstance(gp, DenseGP_GPU):
        check(_lib.mogp_fit_single_GP_MAP(gp._h, int(n_tries), ptr, int(th.size)))
    elif isinstance(gp, MultiOutputGP_GPU):
        check(_lib.mogp_fit_GP_MAP(gp._h, int(n_tries), ptr, int(th.size)))
    else:
        raise TypeError("fit_GP_MAP(): incompatible function arguments")
    return gp


def pivot_cholesky(A):
    """(L, P, rank) = pivoted Cholesky of a symmetric matrix with positive diagonal, computed on the device with the
    semantics of linalg/cholesky.py:284-327 (LAPACK dpstrf; skipped rows get the decreasing replacement diagonal)."""
    A = _f64(np.asarray(A, dtype=np.float64))
    if A.ndim != 2 or A.shape[0] != A.shape[1]:
        raise ValueError("A must have shape (n,n)")
    n = A.shape[0]
    L = np.zeros((n, n))
    P = np.zeros(n, dtype=np.int32)
    rank = ctypes.c_int(0)
    check(_lib.mogp_pivot_cholesky(dptr(A), n, dptr(L), iptr(P), ctypes.byref(rank)))
    return L, P, rank.value


# --------------------------------------------------------------------------------------
# stand-alone kernel objects (bindings.cu:340-361; flat layouts of kernel.hpp:47-107), evaluated on the
# device (mogp_kernel_eval): there is no host implementation of the covariance in the product.
# params = [corr_raw (n_corr), log sigma^2], as the native objects of the reference take them.
# --------------------------------------------------------------------------------------
class _KernelBase(object):
    _kt = kernel_type.SquaredExponential

    def get_n_params(self, inputs):
        return int(np.atleast_2d(inputs).shape[1])

    def _eval(self, what, x1, x2, params):
        x1, x2 = np.atleast_2d(_f64(x1)), np.atleast_2d(_f64(x2))
        if x1.shape[1] != x2.shape[1]:
            raise RuntimeError("kernel inputs must have the same number of columns")
        p = _f64(params, 1)
        n1, D = x1.shape
        n2 = x2.shape[0]
        nc = self.get_n_params(x1)
        out = np.zeros({0: n1 * n2, 1: (nc + 1) * n1 * n2, 2: n2 * n1 * D}[what])
        check(_lib.mogp_kernel_eval(int(self._kt), what, dptr(x1), n1, dptr(x2), n2, D, dptr(p), int(p.size), dptr(out)))
        return out, (n1, n2, D, nc)

    def kernel_f(self, x1, x2, params):
        """sigma^2 k(x1_i, x2_j), shape (n1, n2)"""
        out, (n1, n2, _, _) = self._eval(0, x1, x2, params)
        return out.reshape(n1, n2)

    def kernel_deriv(self, x1, x2, params):
        """d/d theta_p of kernel_f as a flat array of n_params * n1 * n2 entries in (p, i, j) order (the reference returns
        it flat, kernel.hpp:89-107; ``.reshape(n_params, n1, n2)`` is the CPU class's array, Kernel.py:133-173)"""
        return self._eval(1, x1, x2, params)[0]

    def kernel_inputderiv(self, x1, x2, params):
        """d/d x1_i[d] of kernel_f as a flat array of n1 * n2 * D entries in (j, i, d) order (kernel.hpp:68-85, kernel.cu:86-100)"""
        return self._eval(2, x1, x2, params)[0]


class MeanPriors(object):
    """Native-module view of the mean-function priors N(mean, cov) (bindings.cu:558-582, gppriors.hpp): constructed from a
    mean vector and a covariance MATRIX; accessor names of the pybind class.  The numbers the device needs (b, B^-1,
    B^-1 b, log|B|) come from here (DenseGP_GPU.set_mean_priors)."""

    def __init__(self, mean=(), cov=()):
        self._mean = np.reshape(np.array(mean, dtype=np.float64), (-1,))
        self._cov = np.array(cov, dtype=np.float64).reshape(self._mean.size, self._mean.size) if self._mean.size else np.zeros((0, 0))
        if self._mean.size and not np.all(np.diag(self._cov) > 0.):
            raise RuntimeError("all covariances must be greater than zero in MeanPriors")
        self.set_prior_dists()

    def set_prior_dists(self, ptypes=None, priorparams=None):
        """set_prior_dists(): one weak prior per mean parameter; set_prior_dists(types, [[shape, scale], ...]): the given
        distributions -- as in the reference they are only SAMPLED from (start points of a fit), they do not enter the
        log-posterior (gppriors.hpp:244-275)."""
        q = self.get_n_params()
        if ptypes is None:
            self._dists = [WeakPrior() for _ in range(q)]
            return
        ptypes, priorparams = list(ptypes), list(priorparams if priorparams is not None else [])
        if len(ptypes) != q or len(priorparams) != q:
            raise RuntimeError("Number of prior types and parameter sets must equal number of meanfunc parameters.")
        self._dists = [GPPriors.make_prior(t, pp) for t, pp in zip(ptypes, priorparams)]

    def sample(self, transform=None):
        """one draw per mean parameter (gppriors.hpp:277-283); weak priors draw the raw value from U(-2.5, 2.5)"""
        return [d.sample(transform) for d in self._dists]

    def get_mean(self):
        return self._mean.copy()

    def get_cov(self):
        return self._cov.copy()

    def get_n_params(self):
        return int(self._mean.size)

    def has_weak_priors(self):
        return self._mean.size == 0

    def dm_dot_b(self, dm):
        dm = np.asarray(dm, dtype=np.float64)
        return np.zeros(dm.shape[0]) if self.has_weak_priors() else dm @ self._mean

    def inv_cov(self):
        return np.zeros((0, 0)) if self.has_weak_priors() else np.linalg.inv(self._cov)

    def inv_cov_b(self):
        return np.zeros(0) if self.has_weak_priors() else np.linalg.solve(self._cov, self._mean)

    def logdet_cov(self):
        return 0. if self.has_weak_priors() else float(np.linalg.slogdet(self._cov)[1])

    def native_params(self):
        """(q, b, B^-1, B^-1 b, log|B|) for DenseGP_GPU.set_mean_priors"""
        if self.has_weak_priors():
            return 0, np.zeros(1), np.zeros(1), np.zeros(1), 0.
        return (self.get_n_params(), np.ascontiguousarray(self._mean), np.ascontiguousarray(self.inv_cov()),
                np.ascontiguousarray(self.inv_cov_b()), self.logdet_cov())


class SquaredExponentialKernel(_KernelBase):
    _kt = kernel_type.SquaredExponential


class Matern52Kernel(_KernelBase):
    _kt = kernel_type.Matern52


class ProductMat52Kernel(_KernelBase):
    _kt = kernel_type.ProductMat52


class UniformSqExpKernel(_KernelBase):
    _kt = kernel_type.UniformSqExp

    def get_n_params(self, inputs):
        return 1


class UniformMat52Kernel(UniformSqExpKernel):
    _kt = kernel_type.UniformMat52


_KERNEL_OBJECTS = {0: SquaredExponentialKernel, 1: Matern52Kernel, 2: ProductMat52Kernel, 3: UniformSqExpKernel,
                   4: UniformMat52Kernel}
