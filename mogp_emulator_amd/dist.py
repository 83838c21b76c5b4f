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
    if device is None and world > 1:
        device = _collective_device(group)
    if device is not None:
        local_t = local_t.to(device)
    if world == 1:
        return local_t[:n_total]
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
    ``get_indices_not_fit()``: global emulator indices); ``predict`` ends with the single gather of
    means / variances, every rank returns the full (n_emulators, m) arrays.
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
        self.local = factory(inputs, targets[self.lo:self.hi], **kwargs) if self.hi > self.lo else None
        self.fit_ok = np.zeros(self.n_emulators, dtype=bool)
        self.logpost = np.full(self.n_emulators, np.nan)
        self.nuggets = np.full(self.n_emulators, np.nan)
        self.theta_hat = [None] * self.n_emulators

    # -- the fit exchange -----------------------------------------------------------------------------
    def _gather_fit_records(self):
        """ONE collective: rows [fit_ok, logpost, nugget, n_theta, theta_0 .. theta_{P-1}] per emulator."""
        rec = self.local.fit_record() if self.local is not None else {"fit_ok": [], "logpost": [], "nugget": [], "theta": []}
        n_local = len(rec["fit_ok"])
        # fixed record width (equal block shapes on every rank without a second collective): the device kernels take at
        # most 80 inputs, so theta has at most 80 + 2 data and 7 mean entries; 100 doubles per emulator is 50 KB for C3
        width = THETA_PAD
        assert all(t is None or len(t) <= width for t in rec["theta"])
        block = np.zeros((n_local, 4 + width))
        for k in range(n_local):
            th = rec["theta"][k]
            block[k, 0] = 1.0 if rec["fit_ok"][k] else 0.0
            block[k, 1] = rec["logpost"][k]
            block[k, 2] = rec["nugget"][k]
            if th is not None:
                block[k, 3] = len(th)
                block[k, 4:4 + len(th)] = th
        full = gather_rows(block, self.n_emulators, group=self.group).cpu().numpy()
        self.fit_ok = full[:, 0] > 0.5
        self.logpost = np.where(self.fit_ok, full[:, 1], np.nan)
        self.nuggets = full[:, 2].copy()
        self.theta_hat = [full[k, 4:4 + int(full[k, 3])].copy() if self.fit_ok[k] else None for k in range(self.n_emulators)]

    def fit_GP_MAP(self, fit_fn=None, **kwargs):
        if self.local is not None:
            if fit_fn is None:
                from .fitting import fit_GP_MAP as fit_fn
            self.local = fit_fn(self.local, **kwargs)
        self._gather_fit_records()
        return self

    def fit(self, thetas):
        if self.local is not None:
            self.local.fit(np.asarray(thetas)[self.lo:self.hi])
        self._gather_fit_records()

    def get_indices_fit(self):
        return [int(k) for k in np.nonzero(self.fit_ok)[0]]

    def get_indices_not_fit(self):
        return [int(k) for k in np.nonzero(~self.fit_ok)[0]]

    # -- the predict exchange ----------------------------------------------------------------------------
    def predict(self, testing, device=None, **kwargs):
        m = np.atleast_2d(testing).shape[0]
        if self.local is not None:
            mean, unc, _ = self.local.predict(testing, deriv=False, **kwargs)
            payload = np.stack([mean, unc if unc is not None else np.zeros_like(mean)], axis=1)   # (n_local, 2, m)
        else:
            payload = np.zeros((0, 2, m))
        full = gather_rows(payload, self.n_emulators, device=device, group=self.group)
        full = full.cpu().numpy()
        return full[:, 0, :], full[:, 1, :]
