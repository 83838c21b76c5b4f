"""
Multi-GPU execution of the many-emulator axis: ONE process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).  Emulators are independent
(multioutputgp_gpu.hpp:216-228, fitting.hpp:122-127), so they are block-partitioned over ranks with
no data-path collective; the only exchange is a single gather of per-emulator results
(SURVEY.md section 8e).  The reference itself has no multi-device code.
"""
import numpy as np


def shard_bounds(n_items, world_size, rank):
    """Contiguous block [lo, hi) of ceil(n_items / world_size) items owned by ``rank``."""
    per = -(-int(n_items) // int(world_size))
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def shard_sizes(n_items, world_size):
    return [shard_bounds(n_items, world_size, r)[1] - shard_bounds(n_items, world_size, r)[0] for r in range(world_size)]


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
    ``MultiOutputGP_GPU`` on the rank's own device).  fit / predict run on the local shard only;
    ``predict`` ends with the single gather so every rank returns the full (n_emulators, m) arrays.
    """

    def __init__(self, inputs, targets, factory=None, group=None, **kwargs):
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
            from .MultiOutputGP_GPU import MultiOutputGP_GPU
            factory = MultiOutputGP_GPU
        self.local = factory(inputs, targets[self.lo:self.hi], **kwargs) if self.hi > self.lo else None

    def fit_GP_MAP(self, fit_fn=None, **kwargs):
        if self.local is None:
            return self
        if fit_fn is None:
            from .fitting import fit_GP_MAP as fit_fn
        self.local = fit_fn(self.local, **kwargs)
        return self

    def fit(self, thetas):
        if self.local is not None:
            self.local.fit(np.asarray(thetas)[self.lo:self.hi])

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
