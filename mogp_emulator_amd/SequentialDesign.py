"""
MICE candidate scoring -- the second consumer of batched predictions.

``MICEDesign._eval_metric`` (mogp_emulator/SequentialDesign.py:884-964) scores every candidate point c by

    criterion(c) = Var_base[f(c)] / Var_cand\\c[f(c)]

where the denominator is the predictive variance at c of a GP conditioned on all OTHER candidates
(``MICEFastGP.fast_predict``, :705-747; nugget = base nugget * nugget_s).  The reference evaluates the
denominator one candidate at a time, each time re-solving for the full inverse (O(n_cand^3) per
candidate) and downdating it with the Woodbury identity.  The downdated quadratic form has the closed
value 1 / [K^-1]_cc, so here ALL denominators come from one batched device factorisation + one pass
over L^-1 (``loo_variance``), and the numerator from one batched predictive-variance call.

Only the scoring hot path is provided; the sequential-design driver classes (point generation,
simulator binding, bookkeeping) are host control plane and out of scope.
"""
import numpy as np

from .GaussianProcessGPU import GaussianProcessGPU
from .Priors import GPPriors


class MICEFastGP(GaussianProcessGPU):
    """GaussianProcessGPU with ``fast_predict(index)``: leave-one-out predictive variance at a training input."""

    def loo_variance(self):
        if not self.theta.data_has_been_set():
            raise ValueError("hyperparameters have not been fit for this Gaussian Process")
        return self._densegp_gpu.loo_variance()

    def fast_predict(self, index):
        index = int(index)
        assert 0 <= index < self.n, "index must be 0 <= index < n"
        key = tuple(self.theta.get_data())
        if getattr(self, "_loo_key", None) != key:
            self._loo, self._loo_key = self.loo_variance(), key
        return np.array([self._loo[index]])


def mice_criterion(gp, candidates, nugget_s=1.):
    """MICE criterion of every candidate for the fitted base emulator ``gp`` (a GaussianProcessGPU):
    returns (scores (n_cand,), index of the best candidate)."""
    candidates = np.ascontiguousarray(candidates, dtype=np.float64)
    if candidates.ndim == 1:
        candidates = candidates.reshape(-1, 1)
    assert candidates.ndim == 2 and candidates.shape[1] == gp.D, "bad shape for candidates"
    assert nugget_s >= 0., "nugget_s must be non-negative"
    if not gp.theta.data_has_been_set():
        raise ValueError("hyperparameters have not been fit for this Gaussian Process")
    n_cand, D = candidates.shape
    _, unc_base, _ = gp.predict(candidates, unc=True, deriv=False)
    fast = MICEFastGP(candidates, np.ones(n_cand), kernel=gp.kernel, nugget=float(gp.nugget * nugget_s),
                      priors=GPPriors(n_corr=D, nugget_type="fixed"), max_batch_size=max(n_cand, 1))
    fast.fit(np.asarray(gp.theta.get_data())[:D + 1])             # correlation lengths and covariance of the base fit
    scores = unc_base / fast.loo_variance()
    assert np.all(np.isfinite(scores)), "error in computing MICE critera"
    return scores, int(np.argmax(scores))
