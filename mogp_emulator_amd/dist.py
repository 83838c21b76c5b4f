"""
Multi-GPU execution of the many-emulator axis: ONE process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).  Emulators are independent
(multioutputgp_gpu.hpp:216-228, fitting.hpp:122-127), so they are block-partitioned over ranks with
no data-path collective; the only exchange is a single gather of per-emulator results
(SURVEY.md section 8e): after a fit the record (fit_ok, log-posterior, nugget, theta_hat) of every
emulator (fitting.hpp:111-117, MultiOutputGP_GPU.py:316-348), after a prediction the means and
variances.  The reference itself has no multi-device code.
"""
import os

import numpy as np


THETA_PAD = 96      # doubles reserved for theta_hat in a fit record
REC_WIDTH = 5 + THETA_PAD   # [fit_ok, logpost, nugget, len(theta), rank_failed, theta_0 .. theta_95] = 101 doubles per emulator


class ShardError(RuntimeError):
    """The local work of one or more ranks raised; every rank still joined the collective (nobody hangs) and every rank
    raises this error afterwards.  ``ranks``: the ranks that failed (as far as this rank can tell from the gather)."""
    def __init__(self, msg, ranks=()):
        super(ShardError, self).__init__(msg)
        self.ranks = list(ranks)


def shard_bounds(n_items, world_size, rank):
    """Contiguous block [lo, hi) of ceil(n_items / world_size) items owned by ``rank``."""
    per = -(-int(n_items) // int(world_size))
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def shard_sizes(n_items, world_size):
    return [shard_bounds(n_items, world_size, r)[1] - shard_bounds(n_items, world_size, r)[0] for r in range(world_size)]


def _collective_device(group=None):
    """Device the gather payload must live on: the rank's GPU with RCCL ("nccl"), host memory with gloo."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_rows(local, n_total, device=None, group=None):
    """All ranks receive the (n_total, ...) array whose row block [lo, hi) came from each rank.
    ONE all_gather of equal-sized (padded) blocks -- latency bound, payloads are tiny next to xGMI
    bandwidth (64 x m doubles), so no reduction and no ring tuning is involved."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if isinstance(local, np.ndarray):
        local_t = torch.from_numpy(np.ascontiguousarray(local))
    else:
        local_t = local.contiguous()
    if device is None and dist.is_initialized():
        device = _collective_device(group)
    if device is not None:
        local_t = local_t.to(device)
    if not dist.is_initialized():
        return local_t[:n_total]
    # (a process group of ONE rank still goes through the collective: the same RCCL call as with 8 ranks)
    per = -(-int(n_total) // world)
    shape = (per,) + tuple(local_t.shape[1:])
    block = torch.zeros(shape, dtype=local_t.dtype, device=local_t.device)
    block[:local_t.shape[0]] = local_t
    out = torch.empty((world * per,) + tuple(local_t.shape[1:]), dtype=local_t.dtype, device=local_t.device)
    dist.all_gather_into_tensor(out, block, group=group)
    return out[:n_total]


class ShardedMultiOutputGP(object):
    """MultiOutputGP whose emulators are spread over the ranks of a process group.

    ``factory(inputs, local_targets, **kwargs)`` builds the per-rank model (by default
    ``MultiOutputGP_GPU`` on the rank's own device, ``LOCAL_RANK`` modulo the visible devices).
    fit / predict run on the local shard only.  ``fit_GP_MAP`` / ``fit`` end with ONE gather of the
    per-emulator fit records, so that every rank knows theta_hat, the log-posterior, the nugget and the
    fit status of ALL emulators (``theta_hat``, ``logpost``, ``nuggets``, ``get_indices_fit()``,
    ``get_indices_not_fit()``: global emulator indices); ``predict`` has the reference's signature and result
    (MultiOutputGP_GPU.py:185-297: ``unc``, ``deriv``, ``include_nugget``, ``allow_not_fit``) and ends with the single gather of
    means / variances / input derivatives, every rank returns the full arrays.

    Factory protocol (what a custom per-rank model must offer): ``fit(thetas)``, ``predict(testing, unc=, deriv=,
    include_nugget=, allow_not_fit=)`` returning ``(mean, unc, deriv)`` with (n_local, m) / (n_local, m, D) arrays (or None
    for what was not asked for), and -- optional -- ``fit_record()`` returning
    ``{"fit_ok", "logpost", "nugget", "theta"}`` lists of length n_local (``MultiOutputGP_GPU.fit_record``).  A model
    without ``fit_record`` is reported through ``get_indices_fit()`` / ``get_indices_not_fit()`` alone (log-posterior,
    nugget and theta_hat then stay nan / None).  The default model predicts into device buffers (``_mogp_gpu.predict_dev``,
    any mean function) and the gather runs on them directly: one D2H copy of the gathered result per call.

    Failure on one rank: the local work runs inside try / except and the rank ALWAYS joins the collective with an error
    flag in its record (or payload); after the gather every rank raises ``ShardError`` -- a raising rank can therefore
    never leave the others waiting in ``all_gather``.
    """

    def __init__(self, inputs, targets, factory=None, group=None, device_index=None, **kwargs):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        targets = np.asarray(targets, dtype=np.float64)
        if targets.ndim == 1:
            targets = targets.reshape(1, -1)
        self.n_emulators = targets.shape[0]
        self.lo, self.hi = shard_bounds(self.n_emulators, self.world, self.rank)
        if factory is None:
            from . import LibGPGPU
            from .MultiOutputGP_GPU import MultiOutputGP_GPU
            factory = MultiOutputGP_GPU
            # one process per GPU: the engine of this rank lives on its own device
            ndev = max(LibGPGPU.device_count(), 1)
            if device_index is None:
                device_index = int(os.environ.get("LOCAL_RANK", self.rank)) % ndev
            LibGPGPU.set_device(int(device_index))
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.set_device(int(device_index))
            except ImportError:
                pass
            # the default per-rank model predicts into device buffers whatever its mean function is (capi.hip mogp_mogp_predict_dev)
            self._dev_predict = True
        else:
            self._dev_predict = False
        inputs_arr = np.asarray(inputs, dtype=np.float64)
        self.D = 1 if inputs_arr.ndim == 1 else int(inputs_arr.shape[1])
        self._fit_known = False        # fit_ok holds the outcome of a fit gather (every rank: the same values)
        self.local = factory(inputs, targets[self.lo:self.hi], **kwargs) if self.hi > self.lo else None
        self.fit_ok = np.zeros(self.n_emulators, dtype=bool)
        self.logpost = np.full(self.n_emulators, np.nan)
        self.nuggets = np.full(self.n_emulators, np.nan)
        self.theta_hat = [None] * self.n_emulators

    # -- the fit exchange -----------------------------------------------------------------------------
    def _local_record(self):
        if self.local is None:
            return {"fit_ok": [], "logpost": [], "nugget": [], "theta": []}
        if hasattr(self.local, "fit_record"):
            return self.local.fit_record()
        n_local = self.hi - self.lo
        not_fit = set(self.local.get_indices_not_fit()) if hasattr(self.local, "get_indices_not_fit") else set()
        return {"fit_ok": [k not in not_fit for k in range(n_local)], "logpost": [np.nan] * n_local,
                "nugget": [np.nan] * n_local, "theta": [None] * n_local}

    def _gather_fit_records(self, error=None):
        """ONE collective: rows [fit_ok, logpost, nugget, n_theta, rank_failed, theta_0 .. theta_{P-1}] per emulator.
        ``error``: the exception the local fit raised (the rank still joins, with rank_failed = 1 in its rows)."""
        n_local = self.hi - self.lo
        # fixed record width (equal block shapes on every rank without a second collective): the device kernels take at
        # most 80 inputs, so theta has at most 80 + 2 data and 7 mean entries <= THETA_PAD = 96; REC_WIDTH = 101 doubles
        # per emulator is 52 KB for C3
        block = np.zeros((n_local, REC_WIDTH))
        if error is None:
            try:
                rec = self._local_record()
                for k in range(n_local):
                    th = rec["theta"][k]
                    block[k, 0] = 1.0 if rec["fit_ok"][k] else 0.0
                    block[k, 1] = rec["logpost"][k]
                    block[k, 2] = rec["nugget"][k]
                    if th is not None:
                        if len(th) > THETA_PAD:
                            raise ValueError("theta of %d entries does not fit the fit record (%d)" % (len(th), THETA_PAD))
                        block[k, 3] = len(th)
                        block[k, 5:5 + len(th)] = th
            except Exception as exc:           # noqa: BLE001 -- the rank must reach the collective whatever happened
                error = exc
        if error is not None:
            block[:] = 0.
            block[:, 1:3] = np.nan
            block[:, 4] = 1.0
        full = gather_rows(block, self.n_emulators, group=self.group).cpu().numpy()
        self.fit_ok = full[:, 0] > 0.5
        self._fit_known = True
        self.logpost = np.where(self.fit_ok, full[:, 1], np.nan)
        self.nuggets = full[:, 2].copy()
        self.theta_hat = [full[k, 5:5 + int(full[k, 3])].copy() if self.fit_ok[k] and full[k, 3] > 0 else None
                          for k in range(self.n_emulators)]
        self._raise_if_failed(full[:, 4] > 0.5, error, "fit")

    def _raise_if_failed(self, failed_rows, error, what):
        per = -(-self.n_emulators // self.world)
        ranks = sorted(set(int(k) // per for k in np.nonzero(failed_rows)[0]))
        if error is not None and self.rank not in ranks:
            ranks = sorted(ranks + [self.rank])          # a rank without emulators has no row to carry its flag
        if ranks:
            msg = "%s failed on rank(s) %s" % (what, ranks)
            if error is not None:
                msg += "; this rank (%d): %r" % (self.rank, error)
            raise ShardError(msg, ranks) from error

    def fit_GP_MAP(self, fit_fn=None, **kwargs):
        error = None
        if self.local is not None:
            try:
                if fit_fn is None:
                    from .fitting import fit_GP_MAP as fit_fn
                self.local = fit_fn(self.local, **kwargs)
            except Exception as exc:           # noqa: BLE001
                error = exc
        self._gather_fit_records(error)
        return self

    def fit(self, thetas):
        error = None
        if self.local is not None:
            try:
                self.local.fit(np.asarray(thetas)[self.lo:self.hi])
            except Exception as exc:           # noqa: BLE001
                error = exc
        self._gather_fit_records(error)

    def get_indices_fit(self):
        return [int(k) for k in np.nonzero(self.fit_ok)[0]]

    def get_indices_not_fit(self):
        return [int(k) for k in np.nonzero(~self.fit_ok)[0]]

    # -- the predict exchange ----------------------------------------------------------------------------
    def _device_path(self):
        """The default per-rank model with RCCL: predictions stay in HBM until after the gather."""
        import torch.distributed as dist
        return self._dev_predict and dist.is_initialized() and dist.get_backend(self.group) == "nccl"

    def predict(self, testing, unc=True, deriv=True, include_nugget=True, allow_not_fit=False, processes=None, full_cov=False,
                device=None):
        """``MultiOutputGP_GPU.predict`` (MultiOutputGP_GPU.py:185-297) over ALL emulators, on every rank: ``PredictResult`` of
        mean (n_emulators, m), unc (n_emulators, m) -- predictive variance, clipped at 0, + nugget iff include_nugget -- and deriv
        (n_emulators, m, D); arrays that were not asked for are zeros, as in the reference.  ``allow_not_fit``: rows of
        emulators that are not fit are NaN (on every rank) instead of a ``ValueError``.  ONE collective carries everything that
        was asked for: per emulator ``[mean | unc | deriv | error flag]``.  With the nccl backend and the default per-rank model
        the local predictions are written into device buffers (``_mogp_gpu.predict_dev``, any mean function), gathered there by
        RCCL and copied to the host once; otherwise host arrays are gathered (gloo / custom factories).  ``full_cov`` is refused:
        an (m, m) covariance per emulator is not something the single gather is sized for -- ask the per-rank model
        (``.local.predict(..., full_cov=True)``) for the emulators ``[lo, hi)`` it holds."""
        from .GaussianProcessGPU import PredictResult
        if full_cov:
            raise NotImplementedError("ShardedMultiOutputGP.predict(full_cov=True) is not supported: the (n_emulators, m, m) "
                                      "covariances are not gathered across ranks; use .local.predict(..., full_cov=True) for "
                                      "the emulators [%d, %d) of this rank" % (self.lo, self.hi))
        testing = np.array(testing, dtype=np.float64)
        if testing.ndim == 1:
            testing = testing.reshape(-1, 1) if self.D == 1 else testing.reshape(1, -1)
        assert testing.ndim == 2, "testing must be a 2D array"
        assert testing.shape[1] == self.D, "second dimension of testing must be the same as the number of input parameters"
        testing = np.ascontiguousarray(testing)
        # Which emulators are not fit is read from the per-rank models NOW and travels in the gathered rows (flag column = 2), not from
        # the outcome of the last fit gather: a caller may have refitted through `.local` since, and a ValueError raised on
        # some ranks only, in front of the collective, would leave the others waiting in it.  Every rank enters the collective.
        m, D = testing.shape
        n_local = self.hi - self.lo
        # columns of one emulator's row in the single gather
        o_unc = m
        o_der = o_unc + (m if unc else 0)
        o_flag = o_der + (m * D if deriv else 0)
        width = o_flag + 1
        error = None
        nf_local = sorted(self.local.get_indices_not_fit()) if (self.local is not None and hasattr(self.local, "get_indices_not_fit")) else []
        if device is None and self._device_path():
            import torch
            dev = _collective_device(self.group)
            payload = torch.zeros((n_local, width), dtype=torch.float64, device=dev)
            try:
                if self.local is not None:
                    d_x = torch.from_numpy(testing).to(dev)
                    d_mean = torch.empty((n_local, m), dtype=torch.float64, device=dev)
                    d_var = torch.empty((n_local, m), dtype=torch.float64, device=dev) if unc else None
                    d_der = torch.empty((n_local, m * D), dtype=torch.float64, device=dev) if deriv else None
                    self.local._mogp_gpu.predict_dev(d_x.data_ptr(), m, d_mean.data_ptr(), d_var.data_ptr() if unc else None,
                                                     d_der.data_ptr() if deriv else None)
                    payload[:, :m] = d_mean
                    if unc:
                        if include_nugget:
                            d_var += torch.from_numpy(np.asarray(self.local._nuggets())).to(dev)[:, None]
                        payload[:, o_unc:o_der] = torch.clamp_min(d_var, 0.)
                    if deriv:
                        payload[:, o_der:o_flag] = d_der
                    if nf_local:               # (their rows came back as NaN from predict_dev)
                        payload[nf_local, o_flag] = 2.
            except Exception as exc:           # noqa: BLE001
                error = exc
                payload.zero_()
                payload[:, o_flag] = 1.
            full = gather_rows(payload, self.n_emulators, device=dev, group=self.group).cpu().numpy()
        else:
            payload = np.zeros((n_local, width))
            try:
                if self.local is not None:
                    try:
                        res = self.local.predict(testing, unc=unc, deriv=deriv, include_nugget=include_nugget,
                                                 allow_not_fit=allow_not_fit or bool(nf_local))
                    except TypeError:
                        # a custom factory with the protocol of rounds 1-4: predict(testing, deriv=, include_nugget=) -> (mean, unc[, deriv])
                        res = tuple(self.local.predict(testing, deriv=deriv, include_nugget=include_nugget)) + (None, None)
                    payload[:, :m] = res[0]
                    if unc and res[1] is not None:
                        payload[:, o_unc:o_der] = res[1]
                    if deriv and res[2] is not None:
                        payload[:, o_der:o_flag] = np.asarray(res[2]).reshape(n_local, m * D)
                    if nf_local:
                        payload[nf_local, o_flag] = 2.
            except Exception as exc:           # noqa: BLE001
                error = exc
                payload[:] = 0.
                payload[:, o_flag] = 1.
            full = gather_rows(payload, self.n_emulators, device=device, group=self.group).cpu().numpy()
        flag = full[:, o_flag]
        self._raise_if_failed((flag > 0.5) & (flag < 1.5), error, "predict")
        not_fit = flag > 1.5                   # the same array on every rank: it came out of the collective
        if not_fit.any() and not allow_not_fit:
            raise ValueError("Hyperparameters have not been fit for this Gaussian Process")
        means = full[:, :m].copy()
        uncs = full[:, o_unc:o_der].copy() if unc else np.zeros((self.n_emulators, m))
        derivs = full[:, o_der:o_flag].reshape(self.n_emulators, m, D).copy() if deriv else np.zeros((self.n_emulators, m, D))
        if not_fit.any():
            # emulators their rank reported as not fit: NaN rows on every rank, whatever the per-rank model returned
            means[not_fit] = np.nan
            uncs[not_fit] = np.nan
            derivs[not_fit] = np.nan
        return PredictResult(mean=means, unc=uncs, deriv=derivs)

    def __call__(self, testing, processes=None):
        return self.predict(testing, unc=False, deriv=False, processes=processes)[0]
