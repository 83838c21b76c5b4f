"""
Prior distributions on the GP hyper-parameters and the default-prior construction that the
GPU wrappers run on the host before handing (type, shape, scale) triples to the native
backend (reference flow: GaussianProcessGPU.py:143-205, 318-333 -> densegp_gpu.hpp:214-232).

Behavioural reference: mogp_emulator/Priors.py -- GPPriors.default_priors (:85-152),
PriorDist.default_prior (:697-744), default_prior_corr (:746-779), InvGammaPrior.default_prior_mode
(:1012-1064), default_prior_nugget (:1083-1104), min/max spacing (:1151-1188).
"""
import numpy as np
import scipy.stats
from scipy.optimize import root

from .libgpgpu import GammaPrior as _NativeGamma
from .libgpgpu import InvGammaPrior as _NativeInvGamma
from .libgpgpu import LogNormalPrior as _NativeLogNormal
from .libgpgpu import WeakPrior

NUGGET_TYPES = ("fit", "adaptive", "fixed", "pivot")


def input_spacing(column):
    """(median neighbour spacing, total range) of the unique values of one input column;
    zeros when there are too few distinct values."""
    u = np.unique(np.asarray(column, dtype=np.float64).ravel())
    lo = float(np.median(np.diff(u))) if u.size > 2 else 0.
    hi = float(u[-1] - u[0]) if u.size > 1 else 0.
    return lo, hi


class PriorDist(WeakPrior):
    """Marker base for the proper (non-weak) distributions."""
    _frozen = None

    @classmethod
    def default_prior(cls, min_val, max_val):
        """Two-parameter fit putting 0.5 % of the mass below ``min_val`` and 0.5 % above ``max_val``
        (root find in log-parameter space from (0, 0)); WeakPrior() if the solver fails."""
        assert 0. < min_val < max_val, "need 0 < min_val < max_val"
        family = cls._frozen

        def mismatch(logp):
            cdf = family(np.exp(logp[0]), scale=np.exp(logp[1])).cdf
            return np.array([cdf(min_val) - 0.005, cdf(max_val) - 0.995])

        sol = root(mismatch, np.zeros(2))
        if not sol["success"]:
            print("Prior solver failed to converge")
            return WeakPrior()
        return cls(float(np.exp(sol["x"][0])), float(np.exp(sol["x"][1])))

    @classmethod
    def default_prior_corr(cls, column):
        lo, hi = input_spacing(column)
        if lo == 0. or hi == 0.:
            print("Too few unique inputs; defaulting to flat priors")
            return WeakPrior()
        return cls.default_prior(lo, hi)


class InvGammaPrior(_NativeInvGamma, PriorDist):
    _frozen = staticmethod(scipy.stats.invgamma)

    @classmethod
    def default_prior_mode(cls, min_val, max_val):
        """Fallback fit: mode at the geometric mean of the bounds, 99.5 % of the mass below max."""
        assert 0. < min_val < max_val
        mode = np.sqrt(min_val * max_val)

        def mismatch(loga):
            a = np.exp(loga)
            return scipy.stats.invgamma(a, scale=(1. + a) * mode).cdf(max_val) - 0.995

        sol = root(mismatch, 0.)
        if not sol["success"]:
            print("Prior solver failed to converge")
            return WeakPrior()
        a = float(np.exp(sol["x"]).ravel()[0])
        return cls(a, (1. + a) * mode)

    @classmethod
    def default_prior_corr_mode(cls, column):
        lo, hi = input_spacing(column)
        if lo == 0. or hi == 0.:
            print("Too few unique inputs; defaulting to flat priors")
            return WeakPrior()
        return cls.default_prior_mode(lo, hi)

    @classmethod
    def default_prior_nugget(cls, min_val=1.e-8, max_val=1.e-6):
        return cls.default_prior_mode(min_val, max_val)


class GammaPrior(_NativeGamma, PriorDist):
    _frozen = staticmethod(scipy.stats.gamma)


class LogNormalPrior(_NativeLogNormal, PriorDist):
    _frozen = staticmethod(scipy.stats.lognorm)


_FAMILIES = {"invgamma": InvGammaPrior, "gamma": GammaPrior, "lognormal": LogNormalPrior}


class MeanPriors(object):
    """Multivariate-normal prior N(mean, cov) on the coefficients of an analytic mean function: the value object
    of mogp_emulator/Priors.py:423-581 (``cov`` a positive scalar, a vector of variances or a covariance matrix;
    both ``None`` = weak prior).  The device only ever sees b, B^-1, B^-1 b and log|B| (``native_params``)."""

    def __init__(self, mean=None, cov=None):
        if mean is None:
            if cov is not None:
                import warnings
                warnings.warn("Both mean and cov need to be set to form a valid nontrivial MeanPriors object. mean is not "
                              "provided, so ignoring the provided cov.")
            self.mean = self.cov = None
            return
        if cov is None:
            raise ValueError("Both mean and cov need to be set to form a valid MeanPriors object")
        self.mean = np.reshape(np.array(mean, dtype=np.float64), (-1,))
        self.cov = np.array(cov, dtype=np.float64)
        q = len(self.mean)
        if self.cov.ndim == 0:
            assert self.cov > 0., "covariance term must be greater than zero in MeanPriors"
        elif self.cov.ndim == 1:
            assert len(self.cov) == q, "mean and variances must have the same length in MeanPriors"
            assert np.all(self.cov > 0.), "all variances must be greater than zero in MeanPriors"
        elif self.cov.ndim == 2:
            assert self.cov.shape == (q, q), "mean and covariances must have the same shape in MeanPriors"
            assert np.all(np.diag(self.cov) > 0.), "all covariances must be greater than zero in MeanPriors"
        else:
            raise ValueError("Bad shape for the covariance in MeanPriors")

    n_params = property(lambda self: 0 if self.mean is None else len(self.mean))
    has_weak_priors = property(lambda self: self.mean is None)

    def _full_cov(self):
        q = len(self.mean)
        return self.cov if self.cov.ndim == 2 else np.diag(np.broadcast_to(self.cov, (q,)))

    def dm_dot_b(self, dm):
        dm = np.asarray(dm)
        return np.zeros(dm.shape[0]) if self.mean is None else np.dot(dm, self.mean)

    def inv_cov(self):
        return 0. if self.mean is None else np.linalg.inv(self._full_cov())

    def inv_cov_b(self):
        return 0. if self.mean is None else np.linalg.solve(self._full_cov(), self.mean)

    def logdet_cov(self):
        return 0. if self.mean is None else float(np.linalg.slogdet(self._full_cov())[1])

    def native_params(self):
        """(q, b, B^-1 row-major, B^-1 b, log|B|) for ``mogp_densegp_set_mean_priors``; q = 0 for weak priors"""
        if self.mean is None:
            return 0, np.zeros(1), np.zeros(1), np.zeros(1), 0.
        return (len(self.mean), np.ascontiguousarray(self.mean), np.ascontiguousarray(self.inv_cov()),
                np.ascontiguousarray(self.inv_cov_b()), self.logdet_cov())

    def __str__(self):
        return "MeanPriors with mean = {} and cov = {}".format(self.mean, self.cov)


class GPPriors(object):
    """Container: one prior per correlation length, one for the covariance scale, one for the
    nugget (only meaningful when the nugget is fit).  ``None`` entries mean weak priors."""

    def __init__(self, mean=None, corr=None, cov=None, nugget=None, n_corr=None, nugget_type="fit"):
        if corr is None and n_corr is None:
            raise ValueError("Must provide an argument for either corr or n_corr in GPPriors")
        if nugget_type not in NUGGET_TYPES:
            raise AssertionError("Bad value for nugget type in GPPriors")
        # mean-coefficient priors apply to the analytic mean function (analytic_mean=True); None = weak
        if mean is None:
            self.mean = MeanPriors()
        elif isinstance(mean, MeanPriors):
            self.mean = mean
        else:
            try:
                self.mean = MeanPriors(*mean)
            except TypeError:
                raise ValueError("Bad value for defining a MeanPriors object in GPPriors, argument must be an iterable "
                                 "containing the mean vector and the covariance as a float/vector/matrix")
        self._nugget_type = nugget_type
        self.corr = corr if corr is not None else [WeakPrior() for _ in range(int(n_corr))]
        self.cov = cov
        self.nugget = nugget

    # -- properties with the reference's validation ------------------------------------------
    @property
    def corr(self):
        return self._corr

    @corr.setter
    def corr(self, value):
        try:
            value = list(value)
        except TypeError:
            raise TypeError("Correlation priors must be a list of WeakPrior derived objects")
        assert len(value) > 0, "Correlation priors must be a list of nonzero length"
        if not all(isinstance(p, WeakPrior) for p in value):
            raise TypeError("Correlation priors must be a list of WeakPrior derived objects")
        self._corr = value

    @property
    def n_corr(self):
        return len(self._corr)

    @property
    def n_mean(self):
        return self.mean.n_params

    @property
    def cov(self):
        return self._cov

    @cov.setter
    def cov(self, value):
        value = WeakPrior() if value is None else value
        if not isinstance(value, WeakPrior):
            raise TypeError("Covariance prior must be a WeakPrior derived object")
        self._cov = value

    @property
    def nugget_type(self):
        return self._nugget_type

    @property
    def nugget(self):
        return self._nugget

    @nugget.setter
    def nugget(self, value):
        if self._nugget_type != "fit":
            if value is not None:
                print("Nugget type does not support prior distribution, setting to None")
            value = None
        elif value is None:
            value = WeakPrior()
        if not (value is None or isinstance(value, WeakPrior)):
            raise TypeError("Nugget prior must be a WeakPrior derived object or None")
        self._nugget = value

    # -- defaults --------------------------------------------------------------------------------
    @classmethod
    def default_priors(cls, inputs, n_corr, nugget_type="fit", dist="invgamma"):
        """Data-driven defaults: per-dimension correlation-length priors spanning the input
        spacing, InvGamma nugget prior when the nugget is fit, weak covariance prior."""
        if nugget_type not in NUGGET_TYPES:
            raise AssertionError("Bad value for nugget type in GPPriors")
        if isinstance(dist, str):
            try:
                family = _FAMILIES[dist.lower()]
            except KeyError:
                raise TypeError("dist must be a prior distribution to contstruct default priors")
        else:
            family = dist
        inputs = np.asarray(inputs, dtype=np.float64)
        if inputs.ndim == 1:
            inputs = inputs.reshape(-1, 1)
        if inputs.shape[1] == n_corr:
            columns = inputs.T
        elif n_corr == 1:
            columns = inputs.reshape(1, -1)
        else:
            raise ValueError("Number of correlation lengths not compatible with input array")
        corr = []
        for col in columns:
            p = family.default_prior_corr(col)
            if not isinstance(p, family):
                p = InvGammaPrior.default_prior_corr_mode(col)
            corr.append(p)
        nug = InvGammaPrior.default_prior_nugget() if nugget_type == "fit" else None
        return cls(corr=corr, cov=None, nugget=nug, nugget_type=nugget_type)
