#!/usr/bin/env python
"""
bench.py -- headline benchmark of the hot path (BASELINE.json):
GP fits/sec + predict pts/sec, 64 outputs x n=2000 x d=10, fp64, synthetic data, at 1/2/4/8 MI355X.

One process per GPU (launched by torch.distributed.run for N > 1).  The workload is BASELINE's C3: 64 independent
emulators IN TOTAL that share X, block-partitioned over the ranks (mogp_emulator_amd.dist.shard_bounds: 8 per GPU on 8
GPUs) -- "scaling": "strong", the same data whatever N is; the only exchange is ONE RCCL all_gather of the predictive
means / variances at the end of every step.  `--scaling weak` keeps 64 emulators per GPU instead.

A "step" is one pass of the hot path over the rank's shard:
    phase fit      : objective of all emulators at theta  (K build + Cholesky + alpha + logdet + logpost)
    phase fit+grad : objective + gradient (adds L^-1, K^-1, fused gradient reduction)
    phase predict  : mean + variance at m = 10 000 points per emulator (inputs + outputs HBM resident)
`value` = fits/s over the whole job = emulators x steps / time spent in the fit phase (max over ranks);
fit+grad/s and predict pts/s are reported next to it from the same timed steps.

other_configs (N = 1): BASELINE's C2 (one n=2000 emulator, predict m=10^4), C4 (16 x n=5000, Matern-5/2, fitted nugget) and
C5 (n=16000) timed on this GPU with their tagged kernels, each with the CPU oracle beside it (`cpu_baseline`: one / two / one of its outputs fitted
on the host, outside every timed region, within --cpu-budget-s) and its own `parity` block from the values that run produces;
predict_first_call_ms: the first predict after a fit without gradient (it also builds L^-1) next to the steady-state time the headline's
predict rate is taken from; projected_scaling: what 2 / 4 / 8 GPUs would deliver, from the one-GPU time of the shard a rank holds (no
multi-GPU measurement can be made on a one-GPU box); roofline.traffic_source: where `traffic` comes from (a committed PMC pass, not this run); nccl_world1 (N = 1): the two exchange payloads through RCCL in a one-rank process group; shard_sweep also
times fit_GP_MAP with 15 concurrent starts per emulator (the workload users run at shard size).
roofline: for the kernel with the largest share of device time; `achieved` = algorithmic flops (n^3/3 per
Cholesky, m n^2 per predictive variance, ...) of all its launches in the timed steps / their total duration measured with
HIP events recorded on the launch stream inside libmogp_hip.so (mogp_profile_*).  fit_roofline: the whole fit phase
against the fp64 MFMA peak (n^3/3 flops per emulator / phase time).  shard_sweep (N = 1): the per-GPU shard regime of the
multi-GPU runs (8 / 16 / 32 emulators, and 2 x n=5000 of C4) timed on this GPU.  cpu_baseline: the oracle (NumPy/LAPACK
restatement of the reference CPU path) timed on this host on a bounded sample of the same workload; parity_in_bench: the
device results for the emulators the CPU baseline evaluates anyway, compared with it and asserted.
Round 6: compact copies of the blocks above are NESTED in `roofline` (headline_kernel = the one-launch Cholesky behind `value`, fit_phase,
kernels, other_configs, shard_sweep, projected_scaling, fit_GP_MAP) and in `cpu_baseline` (parity_in_bench), because the driver's record
keeps nested values of those two objects but only the names of other top-level keys; for N > 1 kernel times are maxima over the ranks,
`traffic` is null, and rank 0 still times the CPU baseline after the timed region.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TF = 78.6      # AMD MI355X FP64 matrix spec; tools/mfma_probe.hip measures 78.2 on this part
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth(config_id, n, d, n_out, m):
    """SURVEY.md section 8d deterministic synthetic inputs."""
    rng = np.random.default_rng(20240607 + config_id)
    X = rng.uniform(0, 1, (n, d))
    T = np.empty((n_out, n))
    for k in range(n_out):
        w = rng.normal(size=d)
        T[k] = np.sin(2 * np.pi * X @ w / np.sqrt(d)) + 0.1 * (X ** 2) @ np.abs(w) + 0.01 * rng.normal(size=n)
    Xs = rng.uniform(0, 1, (m, d))
    return X, T, Xs


def cpu_baseline(X, T, Xs, theta, nugget):
    """Oracle timed on the host cores, ~10-30 s of CPU work in total:
    (a) BLAS-threaded, one emulator at a time: 3 fits, 1 gradient, predict on 4000 points;
    (b) the reference's own parallel model for many outputs: a process pool over emulators with one BLAS thread
        each (MultiOutputGP / fitting.py:333-335), all emulators of the workload fitted once.
    ``value`` is the better of the two fit rates; ``cores`` the threads / processes that produced it."""
    import multiprocessing as mp
    from oracle import cpu_ref as R
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    fits, values = [], {"logpost": [], "emulators": []}
    for k in range(min(3, T.shape[0])):
        gp = R.GPRef(X, T[k], nugget=nugget)
        t0 = time.perf_counter(); values["logpost"].append(gp.fit(theta)); fits.append(time.perf_counter() - t0)
        values["emulators"].append(k)
    t_fit = float(np.median(fits))
    t0 = time.perf_counter(); values["grad"] = gp.logpost_deriv(theta); t_grad = time.perf_counter() - t0
    ms = min(4000, Xs.shape[0])
    t0 = time.perf_counter(); values["mean"], values["var"], _ = gp.predict(Xs[:ms], include_nugget=False); t_pred = time.perf_counter() - t0
    values["last"] = values["emulators"][-1]
    out = {
        "value": 1.0 / t_fit, "unit": "fits/s", "cores": int(threads), "kind": "port",
        "sample": "BLAS-threaded, n=%d d=%d: fit %.2fs (median of %d emulators), gradient %.2fs, predict %d pts %.2fs (NumPy/LAPACK oracle)" % (
            X.shape[0], X.shape[1], t_fit, len(fits), t_grad, ms, t_pred),
        "fit_grad_per_s": 1.0 / (t_fit + t_grad), "predict_pts_per_s": ms / t_pred,
        "host_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "threaded_fits_per_s": 1.0 / t_fit,
    }
    # (b) process pool, one BLAS thread per worker -- the reference's model for many outputs, Pool(processes=None) = every
    # core (mogp_emulator/fitting.py:298, 333-335).  Two sizes: 32 workers (rounds 1-2) and ALL cores the process may run on,
    # capped by memory (~0.35 GB per worker for the (n, n, d) distance temporary; 1 GB budgeted) -- with more workers than
    # outputs the emulators are cycled so that every worker gets work.  A separate script under a hard timeout, so that a
    # pool that fails to start can never wedge the bench.
    import subprocess
    cores = len(os.sched_getaffinity(0))
    try:
        with open("/proc/meminfo") as fh:
            avail_gb = [int(l.split()[1]) for l in fh if l.startswith("MemAvailable")][0] / 1e6
    except Exception:
        avail_gb = 64.0
    all_workers = int(max(1, min(cores, avail_gb * 0.5 / 1.0)))
    out["pool"] = []
    for workers in sorted(set([max(1, min(T.shape[0], cores // 2, 32)), all_workers])):
        try:
            fits = max(T.shape[0], workers)
            cmd = [sys.executable, os.path.join(ROOT, "oracle", "pool_fit.py"), "2", str(X.shape[0]), str(X.shape[1]),
                   str(T.shape[0]), str(workers), str(fits)]
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            res = json.loads(subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=240, env=env).stdout.strip().splitlines()[-1])
            pool_rate = res["fits"] / res["wall_s"]
            out["pool"].append({"workers": workers, "fits": res["fits"], "wall_s": res["wall_s"], "fits_per_s": pool_rate,
                                "mean_fit_s_per_worker": res["mean_fit_s"]})
            out["sample"] += "; process pool x%d, 1 BLAS thread each: %d fits in %.2fs (%.2fs per fit per worker)" % (
                workers, res["fits"], res["wall_s"], res["mean_fit_s"])
            if workers <= 32:
                out["pool_fits_per_s"], out["pool_workers"] = pool_rate, workers
            else:
                out["pool_all_cores_fits_per_s"], out["pool_all_cores_workers"] = pool_rate, workers
            if pool_rate > out["value"]:
                out["value"], out["cores"] = pool_rate, workers
        except Exception as exc:                                              # the pool is a bonus measurement
            out["pool_error_%d" % workers] = repr(exc)[:200]
    return out, values


def parity_in_bench(mo, values, theta, Xs):
    """The device against the oracle values the CPU baseline has just computed (same X, t, theta, X*): log-posterior of
    three emulators, the full gradient and 4000 predictions of the last one.  Tolerances are the stated ones
    (DESIGN.md section 4): logpost rtol 1e-10, gradient rtol 1e-7 / atol 1e-7 max|g|, mean rtol 1e-7, variance atol 1e-7 sigma^2."""
    B = mo.n_emulators()
    f, g, ok = mo.eval(np.tile(theta, (B, 1)), grad=True)
    assert ok.all()
    k = values["last"]
    ms = values["mean"].shape[0]
    mean, var = np.zeros((B, ms)), np.zeros((B, ms))
    mo.predict_variance_batch(np.ascontiguousarray(Xs[:ms]), mean, var)
    ref_f = np.array(values["logpost"])
    out = {
        "emulators": values["emulators"], "theta": "the fixed benchmark theta", "predict_points": int(ms),
        "max_rel_logpost": float(np.max(np.abs(f[values["emulators"]] - ref_f) / np.abs(ref_f))),
        "max_rel_grad": float(np.max(np.abs(g[k] - values["grad"])) / np.max(np.abs(values["grad"]))),
        "max_rel_mean": float(np.max(np.abs(mean[k] - values["mean"]) / np.maximum(np.abs(values["mean"]), 1e-3))),
        "max_abs_var": float(np.max(np.abs(np.maximum(var[k], 0.) - values["var"]))),
        "tolerances": {"logpost_rtol": 1e-10, "grad_rel_to_max": 1e-7, "mean_rtol": 1e-7, "var_atol": 1e-7},
    }
    assert out["max_rel_logpost"] <= 1e-10 and out["max_rel_grad"] <= 1e-7 and out["max_abs_var"] <= 1e-7, out
    assert np.allclose(mean[k], values["mean"], rtol=1e-7, atol=1e-8), out
    out["passed"] = True
    return out


def time_shard(M, GPPriors, cid, n, d, B, m, kernel, nugget, theta, reps, map_starts=0, map_iters=10):
    """fit / fit+grad / predict time of ONE per-GPU shard of the multi-GPU workload (median of `reps` evaluations)."""
    X, T, Xs = synth(cid, n, d, B, m)
    nt = nugget if isinstance(nugget, str) else "fixed"
    gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=nugget, priors=GPPriors(n_corr=d, nugget_type=nt))
    mo = gp._mogp_gpu
    th = np.tile(theta, (B, 1))
    means, vars_ = np.zeros((B, m)), np.zeros((B, m))

    def med(fn):
        fn(0); fn(1)
        ts = []
        for it in range(reps):
            t0 = time.perf_counter(); fn(it + 2); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3
    t_fit = med(lambda it: mo.eval(th + 1e-3 * it, grad=False))
    t_fg = med(lambda it: mo.eval(th + 1e-3 * it, grad=True))
    t_pr = med(lambda it: mo.predict_variance_batch(Xs, means, vars_))
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    d_Xs = torch.from_numpy(Xs).to(dev)
    d_mean = torch.empty((B, m), dtype=torch.float64, device=dev)
    d_var = torch.empty((B, m), dtype=torch.float64, device=dev)
    t_pr_dev = med(lambda it: mo.predict_variance_batch_dev(d_Xs.data_ptr(), m, d_mean.data_ptr(), d_var.data_ptr()))
    # the same fit through the multi-launch Cholesky schedule of this regime (rounds 1-2), for comparison
    from mogp_emulator_amd import _capi
    _capi.load().mogp_profile_schedule(5, 0)
    t_fit_ml = med(lambda it: mo.eval(th + 1e-3 * it, grad=False))
    _capi.load().mogp_profile_schedule(-1, 0)
    out = {"fit_ms_multi_launch_schedule": t_fit_ml, "n": n, "d": d, "emulators": B, "m": m, "kernel": kernel, "fit_ms": t_fit, "fit_grad_ms": t_fg, "predict_ms_host_buffers": t_pr,
           "predict_ms": t_pr_dev, "fit_ms_per_emulator": t_fit / B, "fit_TFLOPs": B * float(n) ** 3 / 3. / t_fit * 1e-9, "fit_grad_TFLOPs": B * float(n) ** 3 / t_fg * 1e-9}
    if map_starts:
        out.update(time_fit_map(M, X, T, kernel, nugget, map_starts, map_iters))
    return out


def counter(name):
    from mogp_emulator_amd import _capi
    v = ctypes.c_longlong()
    _capi.load().mogp_profile_counter(name.encode(), ctypes.byref(v))
    return int(v.value)


def time_fit_map(M, X, T, kernel, nugget, n_tries, max_iter):
    """The workload users run at shard size: fit_GP_MAP with `n_tries` random starts (mogp_gpu/src/fitting.hpp:61-128;
    default priors, fixed iteration cap).  The starts of an emulator are independent, so the engine evaluates them
    concurrently on a replica engine: a rank that holds 8 emulators factors 8 x 15 = 120 matrices per optimiser round,
    i.e. it works in the full-batch regime of the Cholesky, not in the 8-matrix regime of a single fit(theta)."""
    from mogp_emulator_amd import libgpgpu
    B = T.shape[0]
    # Twice (same seed, same starts, same evaluations): the first call of a process that needs more device memory than the process has
    # touched before pays for the fresh allocations of its replica slots -- 0 to 1.5 s for 25 GB depending on the box (round 6,
    # tools/fitmap_context.py) --, the second one runs on memory the runtime already holds.  `fit_GP_MAP_s` is the second call,
    # `fit_GP_MAP_first_call_s` the first.
    first = None
    for rep in range(2):
        libgpgpu.set_fit_options(max_iter=max_iter, ftol=1e-9, gtol=1e-6, seed=1)
        gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=nugget)              # default priors
        e0, g0 = counter("objective_evals"), counter("gradient_evals")
        r0, sr0 = counter("pool_rounds"), counter("pool_slot_rounds")
        t0 = time.perf_counter()
        libgpgpu.fit_GP_MAP(gp._mogp_gpu, n_tries)
        dt = time.perf_counter() - t0
        if first is None:
            first = dt
    evals = counter("objective_evals") - e0
    rounds = counter("pool_rounds") - r0
    res = {"fit_GP_MAP_s": dt, "fit_GP_MAP_first_call_s": first, "fit_GP_MAP_n_tries": n_tries, "fit_GP_MAP_max_iter": max_iter,
           "fit_GP_MAP_emulator_fits_per_s": B / dt, "fit_GP_MAP_all_fit": len(gp.get_indices_not_fit()) == 0,
           "fit_GP_MAP_objective_evals": evals, "fit_GP_MAP_gradient_evals": counter("gradient_evals") - g0,
           "fit_GP_MAP_objective_evals_per_s": evals / dt,
           "fit_GP_MAP_pool_rounds": rounds, "fit_GP_MAP_mean_batch": (counter("pool_slot_rounds") - sr0) / max(rounds, 1),
           # n^3 / 3 per objective (Cholesky), n^3 with the gradient (+ L^-1, K^-1); the line search asks for the gradient of a
           # trial point only once its objective passed the sufficient-decrease test
           "fit_GP_MAP_TFLOPs": ((counter("gradient_evals") - g0) * 2.0 / 3.0 + evals / 3.0) * float(X.shape[0]) ** 3 / dt * 1e-12}
    libgpgpu.set_fit_options(max_iter=200, ftol=1e-9, gtol=1e-6, seed=1)
    return res


def cpu_baseline_config(tag, X, T, Xs, theta, kernel, nugget, emus, chunk_rows, ms, dev):
    """the CPU path "reported beside each GPU figure".  The oracle (NumPy / LAPACK restatement of the
    reference CPU path; BLAS-threaded) fits `emus` outputs of one of the other configurations at the benchmark theta and predicts
    `ms` points, OUTSIDE any timed region; the device results of the same outputs (`dev`: logpost, mean, var) give the
    configuration's own parity block.  chunk_rows: the oracle builds distances in row chunks where the faithful (n, n, d)
    temporary would not fit (same per-entry arithmetic, tests/test_oracle_golden.py)."""
    from oracle import cpu_ref as R
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    t_fits, t_pred, err_lp, err_mean, err_var, kappa_eps, mean_rtol = [], [], 0., 0., 0., 0., 1e-7
    for k in emus:
        ref = R.GPRef(X, T[k], kernel=kernel, nugget=nugget, chunk_rows=chunk_rows)
        t0 = time.perf_counter(); lp = ref.fit(theta); t_fits.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); mu, var, _ = ref.predict(Xs[:ms], include_nugget=False); t_pred.append(time.perf_counter() - t0)
        dl = np.diag(ref.L)
        kappa_eps = max(kappa_eps, float((dl.max() / dl.min()) ** 2 * np.finfo(float).eps))
        err_lp = max(err_lp, abs(dev["logpost"][k] - lp) / abs(lp))
        # mean: the suite's bar 1e-9 + 1e-7 |mean|, its relative part scaled with the conditioning where that is large: two backward-stable
        # solves of K alpha = t differ by up to cond(K) eps in alpha; cond_2(K) <= trace / lambda_min <= n (sigma^2 + eta) / eta a priori
        # (C2: 2e9, C4: 5e7, C5: 1.6e10), bar = max(1e-7, 0.1 x that bound x eps) -- 1e-7 for C2 and C4, 3.6e-7 for C5
        eta_k = float(ref.nugget)
        cond_bound = X.shape[0] * (float(np.exp(theta[X.shape[1]])) + eta_k) / max(eta_k, 1e-300)
        mean_rtol = max(1e-7, 0.1 * cond_bound * np.finfo(float).eps)
        err_mean = max(err_mean, float(np.max(np.abs(dev["mean"][k, :ms] - mu) / (1e-9 + mean_rtol * np.abs(mu)))))     # in units of the bar
        err_var = max(err_var, float(np.max(np.abs(np.maximum(dev["var"][k, :ms], 0.) - var))))
        del ref
    # stated bars: logpost 1e-10 -- conditioning-scaled max(1e-10, 32 kappa_L eps) where kappa_L eps > 1e-11
    # (C5: two backward-stable factorisations of the same matrix differ by ~cond eps in the quadratic form) --, mean 1e-7, var 1e-7
    lp_tol = max(1e-10, 32 * kappa_eps) if kappa_eps > 1e-11 else 1e-10
    parity = {"emulators": list(emus), "predict_points": int(ms), "max_rel_logpost": float(err_lp), "max_mean_err_over_bar": err_mean,
              "max_abs_var": err_var, "kappa_L_eps": kappa_eps,
              "tolerances": {"logpost_rtol": lp_tol, "mean_bar": "1e-9 + max(1e-7, 0.1 n (sigma^2 + eta) / eta eps) |mean|", "mean_rtol": mean_rtol, "var_atol": 1e-7}}
    parity["passed"] = bool(err_lp <= lp_tol and err_mean <= 1.0 and err_var <= 1e-7)
    t_fit = float(np.median(t_fits))
    return {"value": 1.0 / t_fit, "unit": "fits/s", "cores": int(threads), "kind": "port",
            "sample": "%s: BLAS-threaded oracle, %d of the configuration's outputs: fit %.2fs each (median), predict %d pts %.2fs%s" % (
                tag, len(emus), t_fit, ms, float(np.median(t_pred)), ", distances in %d-row chunks" % chunk_rows if chunk_rows else ""),
            "predict_pts_per_s": ms / float(np.median(t_pred)), "host_cpus": os.cpu_count()}, parity


def time_other_config(M, GPPriors, lib, read_kernels, tag, cid, n, d, B, m, kernel, nugget, theta, reps, cpu=None):
    """One of BASELINE's other configurations (C2, C4, C5) on this GPU: fit / fit+gradient / predict at the fixed theta of
    SURVEY 8d, with the tagged kernels of the fit phase from HIP events (work per unit: SURVEY 8d table).  cpu = (emulators,
    chunk_rows, points, budget): the oracle timed beside it (cpu_baseline_config) unless the budget (seconds left) is used up."""
    X, T, Xs = synth(cid, n, d, B, m)
    nt = nugget if isinstance(nugget, str) else "fixed"
    gp = M.MultiOutputGP_GPU(X, T, kernel=kernel, nugget=nugget, priors=GPPriors(n_corr=d, nugget_type=nt))
    mo = gp._mogp_gpu
    th = np.tile(theta, (B, 1))
    means, vars_ = np.zeros((B, m)), np.zeros((B, m))

    def med(fn, r):
        fn(0)
        ts = []
        for it in range(r):
            t0 = time.perf_counter(); fn(it + 1); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3
    t_fit = med(lambda it: mo.eval(th + 1e-3 * it, grad=False), reps)
    lib.mogp_profile_reset(); lib.mogp_profile_enable(1)
    for it in range(2):
        f, _, ok = mo.eval(th + 1e-3 * (it + 7), grad=False)
    lib.mogp_profile_enable(0)
    kern = read_kernels()
    for v in kern.values():
        v["ms_per_fit"] = v["ms_total"] / 2
    t_fg = med(lambda it: mo.eval(th + 1e-3 * it, grad=True), max(1, reps - 1))
    t_pr_host = med(lambda it: mo.predict_variance_batch(Xs, means, vars_), max(1, reps - 1))
    # X* and the outputs resident in HBM, as in the headline's predict phase (the host-buffer call adds PCIe both ways)
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    d_Xs = torch.from_numpy(Xs).to(dev)
    d_mean = torch.empty((B, m), dtype=torch.float64, device=dev)
    d_var = torch.empty((B, m), dtype=torch.float64, device=dev)
    t_pr = med(lambda it: mo.predict_variance_batch_dev(d_Xs.data_ptr(), m, d_mean.data_ptr(), d_var.data_ptr()), max(1, reps - 1))
    assert ok.all() and np.all(np.isfinite(means)) and np.all(np.isfinite(vars_))
    assert np.allclose(d_mean.cpu().numpy(), means, rtol=1e-12, atol=1e-12)
    fit_tf, fg_tf, pv_tf = B * float(n) ** 3 / 3. / t_fit * 1e-9, B * float(n) ** 3 / t_fg * 1e-9, B * float(m) * float(n) ** 2 / t_pr * 1e-9
    cpu_out = {}
    if cpu is not None:
        emus, chunk_rows, ms, budget = cpu
        if budget["left"] < budget["need"].get(tag, 0.):
            cpu_out["cpu_baseline"] = {"skipped": "CPU time budget of this run used up (%.0f s left, ~%.0f s needed)" % (
                budget["left"], budget["need"].get(tag, 0.))}
        else:
            t0 = time.perf_counter()
            fdev, _, okd = mo.eval(th, grad=False)          # the benchmark theta itself (the timed evaluations perturb it)
            mo.predict_variance_batch(Xs, means, vars_)
            assert okd.all()
            cpu_out["cpu_baseline"], cpu_out["parity"] = cpu_baseline_config(
                tag, X, T, Xs, theta, kernel, nugget, emus, chunk_rows, ms, {"logpost": fdev, "mean": means, "var": vars_})
            cpu_out["cpu_baseline"]["gpu_over_cpu_fit"] = (B / t_fit * 1e3) / cpu_out["cpu_baseline"]["value"]
            budget["left"] -= time.perf_counter() - t0
    return {**cpu_out, "config": tag, "workload": "%d outputs x n=%d x d=%d, %s, nugget %s, predict m=%d" % (B, n, d, kernel, nugget, m),
            "fit_ms": t_fit, "fit_grad_ms": t_fg, "predict_ms": t_pr, "predict_ms_host_buffers": t_pr_host,
            "fits_per_s": B / t_fit * 1e3, "fit_grad_per_s": B / t_fg * 1e3, "predict_pts_per_s": B * m / t_pr * 1e3,
            "fit_TFLOPs": fit_tf, "fit_frac_of_fp64_mfma_peak": fit_tf / FP64_MFMA_PEAK_TF,
            "fit_grad_TFLOPs": fg_tf, "fit_grad_frac": fg_tf / FP64_MFMA_PEAK_TF,
            "predict_TFLOPs_m_n2": pv_tf, "predict_frac": pv_tf / FP64_MFMA_PEAK_TF,
            "flops_per_cholesky": float(n) ** 3 / 3., "kernels_fit": kern, "logpost_checksum": float(np.sum(f))}


def nccl_world1(mo, B, m, dev):
    """RCCL on the one GPU this run has: a process group of ONE rank with the nccl backend, and the two real exchange
    payloads of the multi-GPU path -- the fit records (B x 101 doubles) and the predictions (B x 2 x m doubles, device
    resident) -- through dist.gather_rows -> all_gather_into_tensor.  Asserts the round trip is bit-exact and reports the
    latency.  (With 8 ranks the same call moves 8 such blocks over xGMI.)"""
    import socket
    import torch
    import torch.distributed as dist
    from mogp_emulator_amd.dist import gather_rows, REC_WIDTH
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = {}
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        rec = torch.randn((B, REC_WIDTH), dtype=torch.float64, device=dev)
        pred = torch.randn((B, 2, m), dtype=torch.float64, device=dev)
        for name, payload in (("fit_records", rec), ("predictions", pred)):
            g = gather_rows(payload, B, device=dev)                    # first call: communicator set-up
            torch.cuda.synchronize()
            ts = []
            for _ in range(20):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g = gather_rows(payload, B, device=dev)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            assert g.is_cuda and torch.equal(g, payload), name
            out["gather_us_nccl_world1_" + name] = float(np.median(ts)) * 1e6
            out["gather_bytes_" + name] = int(payload.numel() * 8)
        out["gather_us_nccl_world1"] = out["gather_us_nccl_world1_predictions"]
        out["bit_exact"] = True
        out["backend"] = dist.get_backend()
    except Exception as exc:                                            # reported, never fatal for the headline numbers
        out["error"] = repr(exc)[:300]
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    return out


def nest_summaries(out, n, total_emus, world):
    """The driver's record keeps the contract keys plus `roofline` and `cpu_baseline` with everything nested in them, and only the
    NAMES of the other top-level keys.  So the figures a reader needs beside the dominant kernel's go INSIDE
    `roofline` as compact blocks: the kernel behind `value` (`headline_kernel`: the one-launch Cholesky), the fit phase as a whole,
    every tagged kernel, the other BASELINE configurations, the per-rank shards with the projected scaling, fit_GP_MAP; and the parity
    verdict inside `cpu_baseline`.  The full blocks stay at top level of the line as before."""
    rf = out.get("roofline")
    if not rf:
        return
    kern = out.get("kernels") or {}
    peak_of = {"mfma": FP64_MFMA_PEAK_TF, "latency": FP64_MFMA_PEAK_TF, "hbm": HBM_PEAK_GBS}
    rf["kernels"] = {k: {"bound": v["bound"], "avg_ms": round(v["avg_ms"], 4), "achieved": round(v["achieved"], 2), "unit": v["unit"],
                         "frac": round(v["achieved"] / peak_of[v["bound"]], 4)} for k, v in kern.items()}
    if "mchol" in kern:
        v = kern["mchol"]
        rf["headline_kernel"] = {"kernel": "mchol", "role": "the one-launch Cholesky: the kernel that determines `value` (fits/s)", "bound": "mfma",
                                 "achieved": v["achieved"], "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": v["achieved"] / FP64_MFMA_PEAK_TF,
                                 "avg_launch_ms": v["avg_ms"], "launches": v["launches"], "flops_per_launch": out["config"]["outputs_per_gpu"] * float(n) ** 3 / 3.}
    rf["fit_phase"] = out.get("fit_roofline")
    rf["phase_ms_per_step"] = out.get("phase_ms_per_step")
    rf["fit_grad_per_s"], rf["predict_pts_per_s"] = out.get("fit_grad_per_s"), out.get("predict_pts_per_s")
    if world > 1:
        rf["traffic_note"] = "traffic is null for N > 1 (PMC passes exist for the one-GPU workload only); kernel times are maxima over ranks"
    oc = {}
    for c in out.get("other_configs", []) or []:
        kf = c.get("kernels_fit", {})
        oc[c["config"]] = {"fit_ms": round(c["fit_ms"], 4), "fit_frac": round(c["fit_frac_of_fp64_mfma_peak"], 4), "fit_grad_ms": round(c["fit_grad_ms"], 4),
                           "fit_grad_frac": round(c["fit_grad_frac"], 4), "predict_ms": round(c["predict_ms"], 4), "predict_frac": round(c["predict_frac"], 4),
                           "mchol_ms": round(kf["mchol"]["ms_per_fit"], 4) if "mchol" in kf else None,
                           "mchol_frac": round(kf["mchol"]["achieved"] / FP64_MFMA_PEAK_TF, 4) if "mchol" in kf else None,
                           "cov_build_GBs": round(kf["cov_build"]["achieved"], 1) if "cov_build" in kf else None,
                           "parity_passed": c.get("parity", {}).get("passed"), "cpu_fits_per_s": c.get("cpu_baseline", {}).get("value")}
    if oc:
        rf["other_configs"] = oc
    sw = {}
    for e in out.get("shard_sweep", []) or []:
        key = "%dx%d" % (e["emulators"], e["n"])
        sw[key] = {"fit_ms": round(e["fit_ms"], 4), "fit_frac": round(e["fit_TFLOPs"] / FP64_MFMA_PEAK_TF, 4), "fit_grad_ms": round(e["fit_grad_ms"], 4),
                   "predict_ms": round(e["predict_ms"], 4)}
        if "fit_GP_MAP_TFLOPs" in e:
            sw[key]["fit_GP_MAP_TFLOPs"] = round(e["fit_GP_MAP_TFLOPs"], 2)
            sw[key]["fit_GP_MAP_over_fit_grad"] = round(e["fit_GP_MAP_TFLOPs"] / e["fit_grad_TFLOPs"], 3)
    if sw:
        rf["shard_sweep"] = sw
    if out.get("projected_scaling"):
        rf["projected_scaling"] = {"basis": "one-GPU time of the per-rank shard: a projection, not a multi-GPU measurement",
                                   **{N: {k: round(v[k], 4) for k in ("fit_efficiency", "fit_grad_efficiency", "predict_efficiency")}
                                      for N, v in out["projected_scaling"]["n_gpus"].items()}}
    fm = out.get("fit_GP_MAP_15_starts_64_emulators")
    if fm:
        fg = (out.get("fit_roofline") or {}).get("fit_grad_achieved")
        rf["fit_GP_MAP"] = {"workload": "%d emulators x 15 starts, max_iter 10" % total_emus, "s": round(fm["fit_GP_MAP_s"], 4),
                            "first_call_s": round(fm["fit_GP_MAP_first_call_s"], 4), "TFLOPs": round(fm["fit_GP_MAP_TFLOPs"], 2),
                            "objective_evals": fm["fit_GP_MAP_objective_evals"], "gradient_evals": fm["fit_GP_MAP_gradient_evals"],
                            "over_fit_grad_phase": round(fm["fit_GP_MAP_TFLOPs"] / fg, 3) if fg else None}
    if "cpu_baseline" in out and "parity_in_bench" in out:
        out["cpu_baseline"]["parity_in_bench"] = {k: out["parity_in_bench"][k] for k in ("max_rel_logpost", "max_rel_grad", "max_rel_mean", "max_abs_var", "passed")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--outputs", type=int, default=64, help="emulators: in total (strong scaling) / per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: BASELINE's C3, the 64 outputs are sharded over the GPUs; weak: 64 outputs on every GPU")
    ap.add_argument("--no-shard-sweep", action="store_true")
    ap.add_argument("--m", type=int, default=10000, help="prediction points per emulator")
    ap.add_argument("--kernel", default="SquaredExponential")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=240.,
                    help="host seconds the CPU oracle may spend beside the other configurations (C2 / C4 / C5); what does not fit is reported as skipped")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C2 / C4 / C5 block (other_configs)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed steps (+ the single-stream pass): none of the untimed extras (host-buffer predict, deriv, full_cov, "
                         "fit_GP_MAP, pivot, tsunami) -- for kernel traces whose per-kernel averages should be those of the timed steps")
    # (self-launched ranks get the command line through the environment: torch.distributed.run's own parser trips over script options
    # that are prefixes of its own, e.g. --m)
    args = ap.parse_args(json.loads(os.environ["MOGP_BENCH_ARGV"]) if "MOGP_BENCH_ARGV" in os.environ and "WORLD_SIZE" in os.environ else None)

    # `python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU, the launch line the
    # driver documents), rank 0 of the children prints the JSON line.  Under torch.distributed.run WORLD_SIZE is set and
    # has to agree with --gpus.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            import socket
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus)]
            os.environ["MOGP_BENCH_ARGV"] = json.dumps(sys.argv[1:])
            sys.stdout.flush()
            os.execv(sys.executable, cmd)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s (launch with --nproc-per-node equal to --gpus)" % (
            args.gpus, os.environ["WORLD_SIZE"]))

    # The contract is ONE JSON line on stdout.  RCCL (and other native libraries) print banners to the C-level stdout, which
    # is flushed at exit, i.e. AFTER Python's: keep the real stdout aside for the JSON line and send everything else to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # "nccl" is RCCL over xGMI.  MOGP_BENCH_BACKEND=gloo exists only to exercise the N > 1 control flow on a
    # single-GPU box (all ranks then share device 0 and the collectives go through host memory).
    backend = os.environ.get("MOGP_BENCH_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev if backend != "nccl" else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (rank 0 times the CPU baseline after the timed region while the other ranks wait in the final barrier: minutes at most, but
        # beyond the watchdog's default patience when the host is busy)
        import datetime
        patience = datetime.timedelta(minutes=45)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index), timeout=patience)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=patience)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    import mogp_emulator_amd as M
    from mogp_emulator_amd import _capi, libgpgpu
    from mogp_emulator_amd.Priors import GPPriors
    lib = _capi.load()
    libgpgpu.set_device(dev_index)
    assert M.gpu_usable(), "no gfx950 device / library"

    from mogp_emulator_amd.dist import shard_bounds
    n, d, m = args.n, args.d, args.m
    nugget = 1e-6
    if args.scaling == "strong":
        # C3: the SAME 64 outputs whatever N is, contiguous blocks per rank
        total_emus = args.outputs
        X, T_all, Xs = synth(2, n, d, total_emus, m)
        lo, hi = shard_bounds(total_emus, world, rank)
        T = T_all[lo:hi]
        per_rank = shard_bounds(total_emus, world, 0)[1]
    else:
        # every rank gets its own outputs (different seeds) on the same kind of data
        total_emus = args.outputs * world
        X, T, Xs = synth(2 + 1000 * rank, n, d, args.outputs, m)
        T_all = T
        lo, hi = rank * args.outputs, (rank + 1) * args.outputs
        per_rank = args.outputs
    B = T.shape[0]
    assert B > 0, "more GPUs than emulators"
    theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
    thetas = np.tile(theta, (B, 1))

    gp = M.MultiOutputGP_GPU(X, T, kernel=args.kernel, nugget=nugget, priors=GPPriors(n_corr=d, nugget_type="fixed"))
    mo = gp._mogp_gpu
    d_Xs = torch.from_numpy(Xs).to(dev)
    d_mean = torch.empty((B, m), dtype=torch.float64, device=dev)
    d_var = torch.empty((B, m), dtype=torch.float64, device=dev)
    # the single exchange: equal (padded) blocks of per_rank emulators, [mean | var]
    send = torch.zeros((per_rank, 2, m), dtype=torch.float64, device=coll_dev) if world > 1 else None
    gathered = torch.empty((world * per_rank, 2, m), dtype=torch.float64, device=coll_dev) if world > 1 else None
    torch.cuda.synchronize()

    phase = {"fit": 0.0, "fitgrad": 0.0, "predict": 0.0, "gather": 0.0}

    def step(it, timed):
        th = thetas + 1e-3 * np.sin(it + lo + np.arange(B))[:, None]      # new theta every step: nothing can be cached
        t0 = time.perf_counter()
        f, _, ok = mo.eval(th, grad=False)
        t1 = time.perf_counter()
        f2, g, ok2 = mo.eval(th, grad=True)
        t2 = time.perf_counter()
        mo.predict_variance_batch_dev(d_Xs.data_ptr(), m, d_mean.data_ptr(), d_var.data_ptr())
        t3 = time.perf_counter()
        if world > 1:
            send[:B, 0].copy_(d_mean)
            send[:B, 1].copy_(d_var)
            dist.all_gather_into_tensor(gathered, send)
            torch.cuda.synchronize()
        t4 = time.perf_counter()
        assert ok.all() and ok2.all() and np.all(np.isfinite(g))
        if timed:
            phase["fit"] += t1 - t0; phase["fitgrad"] += t2 - t1; phase["predict"] += t3 - t2; phase["gather"] += t4 - t3
        return f

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        lib.mogp_dev_synchronize()

    for it in range(args.warmup):
        step(it, False)
    lib.mogp_profile_reset()
    lib.mogp_profile_enable(1)
    barrier()
    t_start = time.perf_counter()
    for it in range(args.steps):
        f_last = step(args.warmup + it, True)
    barrier()
    elapsed = time.perf_counter() - t_start
    lib.mogp_profile_enable(0)

    # outside the timed region: the reference-style host-buffer predict (H2D of X*, D2H of mean/var over PCIe)
    # and one end-to-end MAP fit (lock-step L-BFGS, theta0 given, 1 start) for orientation
    extras = {}
    if rank == 0 and world == 1 and not args.no_shard_sweep and (n, d, B) == (2000, 10, 64):
        # the per-GPU shards of the 2 / 4 / 8-GPU runs of THIS workload, and of C4 (16 x n=5000 over 8 GPUs), on one GPU
        # (+ what users run at shard size: fit_GP_MAP with 15 concurrent starts and a fixed cap of 10 iterations)
        sweep = [time_shard(M, GPPriors, 2, n, d, b, m, args.kernel, nugget, theta, 7, map_starts=15) for b in (1, 8, 16, 32)]
        sweep.append(time_shard(M, GPPriors, 4, 5000, 20, 2, m, "Matern52", "fit",
                                np.array([-2. * np.log(0.3 * np.sqrt(20))] * 20 + [0., np.log(1e-4)]), 3))
        extras["shard_sweep"] = sweep
        extras["fit_GP_MAP_15_starts_64_emulators"] = time_fit_map(M, X, T, args.kernel, nugget, 15, 10)
    if rank == 0 and world == 1 and not args.no_extras:
        means_h = np.zeros((B, m)); vars_h = np.zeros((B, m))
        mo.predict_variance_batch(Xs, means_h, vars_h)
        t0 = time.perf_counter()
        mo.predict_variance_batch(Xs, means_h, vars_h)
        extras["predict_pts_per_s_host_buffers"] = B * m / (time.perf_counter() - t0)
        assert np.allclose(means_h, d_mean.cpu().numpy(), rtol=1e-12, atol=1e-12)
        # default predict of the GPU wrapper also returns d mean / d x* (deriv=True), SURVEY 8d: reported separately
        derivs_h = np.zeros((B, m, d))
        t0 = time.perf_counter()
        mo.predict_deriv(Xs, derivs_h)
        extras["predict_deriv_pts_per_s_host_buffers"] = B * m / (time.perf_counter() - t0)
        # consumers fused behind the prediction: one score per query point leaves the device
        t0 = time.perf_counter()
        imp = mo.implausibility(Xs, np.zeros(B), np.full(B, 0.01), np.zeros(B), rank=1)
        extras["implausibility_pts_per_s"] = B * m / (time.perf_counter() - t0)
        mfc = min(m, 1024)
        mu_fc, cov_fc = np.zeros((B, mfc)), np.zeros((B, mfc, mfc))
        t0 = time.perf_counter()
        mo.predict_full_cov(Xs[:mfc], mu_fc, cov_fc)
        extras["full_cov_s_%dx%d_pts" % (B, mfc)] = time.perf_counter() - t0
        assert np.all(np.isfinite(imp)) and np.allclose(mu_fc, means_h[:, :mfc], rtol=1e-9, atol=1e-9)
        libgpgpu.set_fit_options(max_iter=30, ftol=1e-9, gtol=1e-6, seed=1)
        t0 = time.perf_counter()
        libgpgpu.fit_GP_MAP(mo, 1, theta)
        extras["fit_GP_MAP_s_64_emulators_1_start_maxiter30"] = time.perf_counter() - t0
        extras["fit_GP_MAP_all_fit"] = len(mo.get_unfitted_indices()) == 0
        # nugget="pivot" (SURVEY 8f row 4): the same 64 fits through the pivoted Cholesky (2000 sequential pivot steps each)
        gpp = M.MultiOutputGP_GPU(X, T, kernel=args.kernel, nugget="pivot", priors=GPPriors(n_corr=d, nugget_type="pivot"))
        gpp._mogp_gpu.eval(thetas, grad=False)
        t0 = time.perf_counter()
        fp, _, okp = gpp._mogp_gpu.eval(thetas, grad=False)
        extras["pivot_fits_per_s"] = B / (time.perf_counter() - t0)
        assert okp.all()
        del gpp
        # the one benchmark the reference publishes a figure for: its tsunami data (210 simulations, 14 inputs, 64 outputs;
        # tests/golden/tsunamidata.npz is the reference's data file), fit_GP_MAP with the default 15 starts --
        # "roughly 1 second per emulator" on one core of a quad-core laptop (benchmarks/benchmark_tsunami.py:9-11)
        try:
            ts = np.load(os.path.join(ROOT, "tests", "golden", "tsunamidata.npz"))
            libgpgpu.set_fit_options(max_iter=200, ftol=1e-9, gtol=1e-6, seed=1)
            best = None
            for _ in range(2):
                gpt = M.MultiOutputGP_GPU(ts["inputs"], ts["targets"])
                t0 = time.perf_counter()
                gpt = M.fit_GP_MAP(gpt)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            extras["tsunami_benchmark"] = {"n_emulators": int(ts["targets"].shape[0]), "n": int(ts["inputs"].shape[0]),
                                           "d": int(ts["inputs"].shape[1]), "n_tries": 15, "fit_GP_MAP_s": best,
                                           "s_per_emulator": best / ts["targets"].shape[0],
                                           "all_fit": gpt.get_indices_not_fit() == [],
                                           "reference_published_s_per_emulator": 1.0}
            del gpt
        except OSError:
            pass

    # The timed predict follows eval(grad=True), i.e. L^-1 exists (steady state: many predictions per fit).  The FIRST predict after a
    # plain fit also builds L^-1 (the reference pays its invQ inside fit, densegp_gpu.hpp:576-582): reported next to it.
    if rank == 0 and world == 1:
        first, steady = [], []
        for it in range(3):
            mo.eval(thetas + 1e-3 * (it + 11), grad=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); mo.predict_variance_batch_dev(d_Xs.data_ptr(), m, d_mean.data_ptr(), d_var.data_ptr()); t1 = time.perf_counter()
            mo.predict_variance_batch_dev(d_Xs.data_ptr(), m, d_mean.data_ptr(), d_var.data_ptr()); t2 = time.perf_counter()
            first.append(t1 - t0); steady.append(t2 - t1)
        extras["predict_first_call_ms"] = float(np.median(first)) * 1e3
        extras["predict_steady_ms"] = float(np.median(steady)) * 1e3
        extras["predict_first_call_note"] = ("first predict after a fit without gradient = steady-state predict + the one-off L^-1 build "
                                             "(%.2f ms for %d emulators); predict_pts_per_s is the steady-state rate" % (
                                                 (float(np.median(first)) - float(np.median(steady))) * 1e3, B))
        extras["predict_pts_per_s_first_call"] = B * m / float(np.median(first))

    # per-kernel device times from HIP events on the launch stream
    def read_kernels():
        kern = {}
        for tag, bound in (("mchol", "mfma"), ("chol_update", "mfma"), ("chol_diag128", "latency"), ("chol_trsm128", "hbm"), ("syrk_trailing", "mfma"),
                           ("trtri_merge", "mfma"), ("kinv", "mfma"), ("predict_var", "mfma"),
                           ("cov_build", "hbm"), ("cross_cov", "hbm"), ("grad_reduce", "hbm")):
            ms, cnt, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
            if lib.mogp_profile_get(tag.encode(), ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by)) == 0 and cnt.value:
                sec = ms.value * 1e-3
                use_flops = bound in ("mfma", "latency")
                kern[tag] = {"bound": bound, "launches": cnt.value, "ms_total": ms.value, "avg_ms": ms.value / cnt.value,
                             "achieved": (fl.value / sec * 1e-12) if use_flops else (by.value / sec * 1e-9),
                             "unit": "TFLOP/s" if use_flops else "GB/s"}
                if bound == "hbm" and fl.value:
                    kern[tag]["TFLOP/s"] = fl.value / sec * 1e-12
        return kern
    kern = read_kernels()
    if world > 1:
        # every rank times its own shard: a kernel's figure in the line is that of the SLOWEST rank (max of the total time over ranks;
        # launches / flops are those of a rank's shard, which has per_rank emulators everywhere but possibly the last rank)
        tags = ["mchol", "chol_update", "chol_diag128", "chol_trsm128", "syrk_trailing", "trtri_merge", "kinv", "predict_var", "cov_build", "cross_cov", "grad_reduce"]
        mine = torch.tensor([kern.get(t, {}).get("ms_total", 0.) for t in tags], dtype=torch.float64, device=coll_dev)
        worst = mine.clone()
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        for t, a, b in zip(tags, mine.cpu().tolist(), worst.cpu().tolist()):
            if t in kern and a > 0.:
                kv = kern[t]
                kv["ms_total_this_rank"] = a
                kv["ms_total"], kv["avg_ms"], kv["achieved"] = b, b / kv["launches"], kv["achieved"] * a / b
                if "TFLOP/s" in kv:
                    kv["TFLOP/s"] *= a / b
                kv["note"] = "time = max over the %d ranks; work = one rank's shard" % world
    # The factorisation overlaps kernels of two streams, so the event time of a Cholesky kernel above includes the share
    # of the device it did NOT have.  Its time alone: the same evaluation serialised onto one stream (outside the timed
    # region, N = 1 only); `achieved` of the chol_* kernels is taken from this pass.
    kern_serial = None
    if rank == 0 and world == 1:
        lib.mogp_profile_reset()
        lib.mogp_profile_schedule(-1, 1)
        mo.eval(thetas, grad=False)
        lib.mogp_profile_enable(1)
        t0 = time.perf_counter()
        for it in range(3):
            mo.eval(thetas + 1e-3 * (it + 1), grad=False)
        serial_ms = (time.perf_counter() - t0) / 3 * 1e3
        lib.mogp_profile_enable(0)
        lib.mogp_profile_schedule(-1, 0)
        kern_serial = {k: v for k, v in read_kernels().items() if k.startswith("chol_") or k in ("cov_build", "syrk_trailing", "mchol")}
        for k, v in kern_serial.items():
            v["ms_per_fit_phase"] = v["ms_total"] / 3
            if k in kern:
                kern[k]["achieved_overlapped"] = kern[k]["achieved"]
                kern[k]["achieved"] = v["achieved"]
                kern[k]["note"] = "achieved = one stream (kernel alone on the device); achieved_overlapped = inside the two-stream schedule"
        kern_serial["fit_ms_single_stream"] = serial_ms
        # BASELINE's other configurations on this GPU (VERDICT r2, row (+)2): C4 = 16 outputs, Matern-5/2 + fitted nugget,
        # n=5000, d=20; C5 = one output, n=16000, d=8.  ~0.5 s of GPU time each.
        if not args.no_other_configs and (n, d, B) == (2000, 10, 64):
            # CPU oracle beside each of them, bounded: ~1 s (C2), ~30 s (two of C4's outputs), ~80 s (C5's fit)
            budget = {"left": 0. if args.no_cpu_baseline else float(args.cpu_budget_s), "need": {"C2": 2., "C4": 45., "C5": 100.}}
            cpu_of = (lambda emus, chunk, pts: None) if args.no_cpu_baseline else (lambda emus, chunk, pts: (emus, chunk, pts, budget))
            extras["other_configs"] = [
                # C2: the single-output fit + 10^4-point predict every GaussianProcessGPU.fit / .predict of the reference
                # wrapper runs (GaussianProcessGPU.py:431-438, densegp_gpu.hpp:451-474): one matrix, chain-bound
                time_other_config(M, GPPriors, lib, read_kernels, "C2", 2, 2000, 10, 1, m, "SquaredExponential", 1e-6,
                                  np.array([-2. * np.log(0.3 * np.sqrt(10))] * 10 + [0.]), 9, cpu=cpu_of([0], None, 2000)),
                time_other_config(M, GPPriors, lib, read_kernels, "C4", 4, 5000, 20, 16, m, "Matern52", "fit",
                                  np.array([-2. * np.log(0.3 * np.sqrt(20))] * 20 + [0., np.log(1e-4)]), 3, cpu=cpu_of([0, 15], 256, 500)),
                time_other_config(M, GPPriors, lib, read_kernels, "C5", 5, 16000, 8, 1, m, "SquaredExponential", 1e-6,
                                  np.array([-2. * np.log(0.3 * np.sqrt(8))] * 8 + [0.]), 3, cpu=cpu_of([0], 128, 256))]
            extras["other_configs_cpu_budget_s"] = {"given": float(args.cpu_budget_s), "left": budget["left"]}
        # RCCL exercised on the single GPU: the exchange payloads of the N > 1 path through a world-size-1 nccl group
        if not dist.is_initialized():
            extras["nccl_world1"] = nccl_world1(mo, B, m, dev)

    # max over ranks of every time
    times = torch.tensor([elapsed, phase["fit"], phase["fitgrad"], phase["predict"], phase["gather"]], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    elapsed, t_fit, t_fg, t_pr, t_ga = [float(x) for x in times.cpu()]

    if rank == 0:
        K = args.steps
        # dominant kernel of the step among the roofline-bound ones (the latency-bound diagonal block has no roofline)
        cands = [k for k in kern if kern[k]["bound"] in ("mfma", "hbm")]
        dom = max(cands, key=lambda k: kern[k]["ms_total"]) if cands else None
        roofline = None
        if dom:
            kd = kern[dom]
            peak = FP64_MFMA_PEAK_TF if kd["bound"] == "mfma" else HBM_PEAK_GBS
            # HBM-side bytes per launch from rocprofv3 PMC passes (cannot be collected from inside this process):
            # measured offline on this exact default workload and stored under profiles/; null for any other workload
            traffic, traffic_source = None, ("null: N > 1 -- PMC passes exist for the one-GPU workload only" if world > 1 else "none: not the default workload")
            try:
                if (n, d, B, m, args.kernel, world) == (2000, 10, 64, 10000, "SquaredExponential", 1):
                    import glob
                    tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
                    with open(tf) as fh:
                        traffic = json.load(fh).get(dom, {}).get("traffic_bytes_per_launch")
                    traffic_source = ("committed file profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this exact workload, "
                                      "collected offline (tools/pmc_fetch.sh) -- NOT measured in this run" % os.path.basename(tf))
            except (OSError, ValueError, IndexError):
                traffic, traffic_source = None, "none: no profiles/r*_traffic.json"
            roofline = {"kernel": dom, "bound": kd["bound"], "achieved": kd["achieved"], "peak": peak, "unit": kd["unit"],
                        "frac": kd["achieved"] / peak, "traffic": traffic, "traffic_source": traffic_source,
                        "avg_launch_ms": kd["avg_ms"], "launches": kd["launches"]}
        # the fit phase as a whole against the fp64 MFMA peak: n^3/3 flops per emulator / phase time
        fit_tf = total_emus * K * float(n) ** 3 / 3. / t_fit * 1e-12
        fitgrad_tf = total_emus * K * float(n) ** 3 / t_fg * 1e-12
        out = {
            "metric": "GP fits/sec (+ fit+grad/s, predict pts/s), %d-output n=%d d=%d on %d MI355X, fp64" % (total_emus, n, d, world),
            "value": total_emus * K / t_fit, "unit": "fits/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "MultiOutputGP %d outputs in total (%d on this GPU) x n=%d x d=%d, %s kernel, fixed nugget 1e-6, predict m=%d (unc=True)" % (
                total_emus, B, n, d, args.kernel, m), "outputs_total": total_emus, "outputs_per_gpu": per_rank, "n": n, "d": d, "m_predict": m,
                "parallelism": "emulator-shard x%d, one all_gather of (mean, var) per step" % world},
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            "collective_backend": dist.get_backend() if dist.is_initialized() else None,
            # one step = fit phase + fit+gradient phase + predict phase (+ gather); the headline metric has two parts
            # (fits/s and predict pts/s), each taken from its own phase of the SAME timed K steps, max over ranks:
            "value_definition": "value = outputs * steps / (time of the fit phases inside the timed steps); "
                                "predict_pts_per_s = outputs * m * steps / (time of the predict + gather phases); "
                                "ms_per_step covers all three phases",
            "fit_grad_per_s": total_emus * K / t_fg,
            "predict_pts_per_s": total_emus * m * K / (t_pr + t_ga),
            "phase_ms_per_step": {"fit": t_fit / K * 1e3, "fit_grad": t_fg / K * 1e3, "predict": t_pr / K * 1e3, "gather": t_ga / K * 1e3},
            "roofline": roofline,
            "fit_roofline": {"bound": "mfma", "achieved": fit_tf, "peak": FP64_MFMA_PEAK_TF * world, "unit": "TFLOP/s",
                             "frac": fit_tf / (FP64_MFMA_PEAK_TF * world), "flops_per_fit": float(n) ** 3 / 3.,
                             "fit_grad_achieved": fitgrad_tf, "fit_grad_frac": fitgrad_tf / (FP64_MFMA_PEAK_TF * world)},
            "kernels": kern, "kernels_fit_single_stream": kern_serial,
            "logpost_checksum": float(np.sum(f_last)),
        }
        out.update(extras)
        for e in out.get("shard_sweep", []):
            if e["n"] == n:      # per-emulator fit time of the shard relative to the full batch on one GPU (2.0 = half the efficiency)
                e["per_emulator_time_vs_full_batch"] = e["fit_ms_per_emulator"] / (t_fit / K * 1e3 / total_emus)
        if world == 1 and out.get("shard_sweep"):
            # No multi-GPU run can be made from a one-GPU box: what N GPUs would deliver under strong scaling, from the time ONE GPU
            # takes for the shard a rank holds at that N (shards are independent; the only exchange is the gather measured in
            # nccl_world1, microseconds).  efficiency = (one-GPU time / N) / shard time.
            by_b = {e["emulators"]: e for e in out["shard_sweep"] if e["n"] == n}
            proj = {}
            for N in (2, 4, 8):
                e = by_b.get(total_emus // N)
                if e:
                    proj[str(N)] = {"emulators_per_gpu": total_emus // N, "fit_ms": e["fit_ms"], "fits_per_s": total_emus / e["fit_ms"] * 1e3,
                                    "fit_efficiency": (t_fit / K * 1e3 / N) / e["fit_ms"],
                                    "fit_grad_efficiency": (t_fg / K * 1e3 / N) / e["fit_grad_ms"],
                                    "predict_pts_per_s": total_emus * m / e["predict_ms"] * 1e3,
                                    "predict_efficiency": (t_pr / K * 1e3 / N) / e["predict_ms"]}
            out["projected_scaling"] = {"basis": "one-GPU time of the per-rank shard (shard_sweep) -- a projection, not a multi-GPU measurement",
                                        "n_gpus": proj}
        if not args.no_cpu_baseline:
            # rank 0, after the timed region, for every N ("in the same run"); the pool fits all outputs of the
            # workload (strong scaling: the 64 of C3), the device side of the parity block is rank 0's shard
            out["cpu_baseline"], values = cpu_baseline(X, T_all if args.scaling == "strong" else T, Xs, theta, nugget)
            out["parity_in_bench"] = parity_in_bench(mo, values, theta, Xs)
        nest_summaries(out, n, total_emus, world)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
