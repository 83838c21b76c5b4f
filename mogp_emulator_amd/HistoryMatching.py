"""
``HistoryMatching`` -- the consumer of batched predictions of mogp_emulator/HistoryMatching.py:5-700,
re-hosted on the device path.

With a GPU emulator and query coordinates the score

    I_k(x) = |z_k - E f_k(x)| / sqrt(Var f_k(x) + discrepancy_k + obsvar_k),   score = (rank+1)-th largest over k

is evaluated by ``libmogp_hip.so`` directly behind the batched prediction: a sweep over m query points
moves m doubles to the host instead of 2 * n_outputs * m.  With explicit ``expectations`` (a
``PredictResult``) the same arithmetic runs in NumPy, as in the reference (HistoryMatching.py:262-276).

Same public surface and error behaviour as the reference class; the validation helpers are
deliberately compact.
"""
import numpy as np

from .GaussianProcessGPU import GaussianProcessGPU, PredictResult
from .MultiOutputGP_GPU import MultiOutputGP_GPU

_GP_TYPES = (GaussianProcessGPU, MultiOutputGP_GPU)


def implausibility_from_expectations(obs, obs_var, mean, var, discrepancy=0., rank=1):
    """NumPy form of HistoryMatching.get_implausibility for given predictions: mean / var (n_obs, m) or (m,)."""
    mean, var = np.atleast_2d(mean), np.atleast_2d(var)
    n_obs = mean.shape[0]
    total = var + np.reshape(np.broadcast_to(np.atleast_1d(discrepancy), (n_obs,)), (-1, 1)) \
        + np.reshape(np.broadcast_to(np.atleast_1d(obs_var), (n_obs,)), (-1, 1))
    scores = np.abs(np.reshape(obs, (-1, 1)) - mean) / np.sqrt(total)
    kth = n_obs - rank - 1
    return np.partition(scores, kth, axis=0)[kth]


class HistoryMatching(object):
    def __init__(self, gp=None, obs=None, coords=None, expectations=None, threshold=3.):
        self.gp = self.obs = self.coords = self.expectations = None
        self.ndim = self.ncoords = self.threshold = None
        self.I = self.NROY = self.RO = None
        for check, setter, value in ((self.check_gp, self.set_gp, gp), (self.check_obs, self.set_obs, obs),
                                     (self.check_coords, self.set_coords, coords),
                                     (self.check_expectations, self.set_expectations, expectations),
                                     (self.check_threshold, self.set_threshold, threshold)):
            if check(value):
                setter(value)
        self.update()

    # -- the computation -----------------------------------------------------------------------------
    def get_n_obs(self):
        return len(self.obs[0])

    def _mode(self):
        have_gp = self.check_coords(self.coords) and self.check_gp(self.gp)
        have_exp = self.check_expectations(self.expectations)
        if have_gp and have_exp:
            raise ValueError("Multiple valid parameter combinations are set. Previously set parameters can be "
                             "removed by setting them to None")
        if not have_gp and not have_exp:
            raise ValueError("Expectations are not provided, nor is a GP and coordinates. Must set one in order "
                             "to perform History Matching")
        if self.ncoords is None:
            raise ValueError("ncoords is not set despite a valid parameter combination being found.")
        return "gp" if have_gp else "expectations"

    def _select_expectations(self):
        """predictions the score is computed from (reference helper, HistoryMatching.py:155-195)"""
        if self._mode() == "gp":
            return self.gp.predict(self.coords)
        return self.expectations

    def get_implausibility(self, discrepancy=0., rank=1):
        if not self.check_obs(self.obs):
            raise ValueError("implausibility calculation requires that the observation value is set. This can be "
                             "done using the set_obs method.")
        assert np.all(np.asarray(discrepancy) >= 0.), "Model discrepancy variance cannot be negative"
        discrepancy = np.atleast_1d(np.asarray(discrepancy, dtype=np.float64))
        n_obs = self.get_n_obs()
        if n_obs == 1:
            rank = 0
        assert rank >= 0, "rank must be a non-negative integer"
        assert rank < n_obs, "rank must be less than the number of observations"
        z = np.asarray(self.obs[0], dtype=np.float64)
        zvar = np.broadcast_to(np.asarray(self.obs[1], dtype=np.float64), z.shape)
        disc = np.broadcast_to(discrepancy, z.shape)
        if self._mode() == "gp" and self._device_path_ok(rank):
            native = self.gp._mogp_gpu if isinstance(self.gp, MultiOutputGP_GPU) else self.gp._densegp_gpu
            n_out = self.gp.n_emulators if isinstance(self.gp, MultiOutputGP_GPU) else 1
            assert n_obs == n_out
            coords = np.ascontiguousarray(self.coords, dtype=np.float64)
            if isinstance(self.gp, MultiOutputGP_GPU):
                self.I = native.implausibility(coords, z, zvar, disc, include_nugget=True, rank=rank)
            else:
                self.I = native.implausibility(coords, z[0], zvar[0], disc[0], include_nugget=True)
            return self.I
        expectations = self._select_expectations()
        assert n_obs == np.atleast_2d(expectations[0]).shape[0]
        assert n_obs == np.atleast_2d(expectations[1]).shape[0]
        self.I = implausibility_from_expectations(z, zvar, expectations[0], expectations[1], disc, rank)
        return self.I

    def _device_path_ok(self, rank):
        """the fused kernel covers zero / fixed mean functions, rank <= 15 and fully fitted emulators"""
        if rank > 15:
            return False
        if isinstance(self.gp, MultiOutputGP_GPU):
            native, fitted = self.gp._mogp_gpu, len(self.gp.get_indices_not_fit()) == 0
            first = native.emulator(0)
        else:
            first, fitted = self.gp._densegp_gpu, self.gp.theta.data_has_been_set()
        return fitted and first.get_theta().get_n_mean() == 0 and first.get_beta().size == 0

    def get_NROY(self, discrepancy=0., rank=1):
        if self.I is None:
            self.get_implausibility(discrepancy, rank)
        self.NROY = list(np.where(self.I <= self.threshold)[0])
        return self.NROY

    def get_RO(self, discrepancy=0., rank=1):
        if self.I is None:
            self.get_implausibility(discrepancy, rank)
        self.RO = list(np.where(self.I > self.threshold)[0])
        return self.RO

    # -- setters ---------------------------------------------------------------------------------------
    def set_gp(self, gp):
        if not self.check_gp(gp):
            raise TypeError("bad input for set_gp - expects a GaussianProcess object.")
        self.gp = gp

    def set_obs(self, obs):
        if not self.check_obs(obs):
            raise TypeError("bad input for set_obs")
        if isinstance(obs, float):
            self.obs = [np.array([obs]), np.array([0.])]
        elif len(obs) == 1:
            self.obs = [np.atleast_1d(obs[0]), np.array([0.])]
        else:
            self.obs = [np.atleast_1d(a) for a in obs]

    def set_coords(self, coords):
        if coords is not None and not self.check_coords(coords):
            raise TypeError("bad input for set_coords - expected coords in the form of a list or 1D or 2D ndarray "
                            "of numerical values")
        self.coords = None if coords is None else (coords.reshape(-1, 1) if coords.ndim == 1 else coords)
        self.update()

    def set_expectations(self, expectations):
        if expectations is not None and not self.check_expectations(expectations):
            raise TypeError("bad input for set_expectations - expected a Tuple of 3 numpy arrays")
        self.expectations = expectations
        self.update()

    def set_threshold(self, threshold):
        if not self.check_threshold(threshold):
            raise TypeError("bad input for set_threshold - expected a float")
        self.threshold = float(threshold)

    # -- validation ------------------------------------------------------------------------------------
    def check_gp(self, gp):
        return isinstance(gp, _GP_TYPES)

    def check_obs(self, obs):
        if obs is None:
            return False
        if isinstance(obs, np.ndarray):
            if obs.ndim > 2:
                raise ValueError("bad input for HistoryMatching, the obs parameter must be at most 2D")
            assert obs.shape[0] == 2, "first dimension of observations must have length 2"
        elif isinstance(obs, (list, tuple)):
            if len(obs) > 2:
                raise ValueError("bad input type for HistoryMatching - the specified observation parameter must "
                                 "contain at most 2 entries (value, variance)")
            try:
                parts = [np.atleast_1d(np.array(a, dtype=np.float64)) for a in obs]
            except (TypeError, ValueError):
                raise TypeError("bad input type for HistoryMatching - the specified observation parameter must "
                                "contain numerical values")
            if len(parts) == 2 and not (len(parts[0]) == len(parts[1]) or len(parts[1]) == 1):
                raise ValueError("Bad input for values to history matching -- observations and variances must "
                                 "have the same length")
        else:
            try:
                float(obs)
            except (TypeError, ValueError):
                raise TypeError("bad input type for HistoryMatching - the specified observation parameter must be "
                                "a float, a list or an array")
            return True
        if len(obs) == 2:
            assert np.all(np.asarray(obs[1]) >= 0.), "variance in observations cannot be negative"
        return True

    def check_coords(self, coords):
        return isinstance(coords, np.ndarray) and coords.ndim <= 2

    def check_expectations(self, expectations):
        if not isinstance(expectations, PredictResult):
            return False
        mean, unc, deriv = expectations
        if not (isinstance(mean, np.ndarray) and isinstance(unc, np.ndarray) and (deriv is None or isinstance(deriv, np.ndarray))):
            raise TypeError("bad input type for HistoryMatching - expected expectation values to be numpy arrays")
        if mean.shape != unc.shape:
            raise ValueError("bad input for HistoryMatching - mean and variance must have the same shape")
        assert np.all(unc >= 0.), "all variances must be non-negative"
        return True

    def check_threshold(self, threshold):
        if threshold is None:
            return False
        try:
            value = float(threshold)
        except TypeError:
            return False
        assert value >= 0., "threshold must be non-negative"
        return True

    def update(self):
        if self.check_coords(self.coords):
            self.ncoords, self.ndim = self.coords.shape
        elif self.check_expectations(self.expectations):
            self.ncoords = self.expectations[0].shape[-1]

    def status(self):
        print(str(self))

    def __str__(self):
        def shape(x):
            return None if x is None else np.shape(x)
        return ("History Matching tools created with:\n"
                "gp: {}\nobs: {}\ncoords: {}\nexpectations: {}\nthreshold: {}\nI: {}\nNROY: {}\nRO: {}\nndim: {}\nncoords: {}"
                .format(type(self.gp).__name__ if self.gp is not None else None, self.obs, shape(self.coords),
                        None if self.expectations is None else shape(self.expectations[0]), self.threshold, shape(self.I),
                        None if self.NROY is None else len(self.NROY), None if self.RO is None else len(self.RO),
                        self.ndim, self.ncoords))
