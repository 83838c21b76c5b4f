"""Two ranks on ONE GPU with the real per-rank model ("1 GPU pretending to be k shards"): the emulator
partition, the single gather of fit records (theta_hat, log-posterior, nugget, fit status) and the single gather of
predictions, compared with the unsharded MultiOutputGP_GPU.  The process group is gloo (both ranks share device 0; RCCL
wants one device per rank), so the payloads cross host memory -- the collective call sites are the ones bench.py uses
with "nccl"."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _data(n_out):
    rng = np.random.default_rng(5)
    n, d = 300, 3
    X = rng.uniform(0, 1, (n, d))
    T = np.stack([np.sin(3 * X @ rng.normal(size=d)) + 0.05 * rng.normal(size=n) for _ in range(n_out)])
    Xs = rng.uniform(0, 1, (40, d))
    return X, T, Xs


def _aborts():
    import ctypes
    from mogp_emulator_amd import _capi
    v = ctypes.c_longlong()
    _capi.load().mogp_profile_counter(b"mchol_aborts", ctypes.byref(v))
    return int(v.value)


def _worker(rank, world, port, n_out, q, chol):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    if chol:
        os.environ["MOGP_CHOL"] = chol        # "left": one multi-launch schedule whatever the shard size, bit-for-bit comparable
    else:
        os.environ.pop("MOGP_CHOL", None)     # default: the one-launch Cholesky -- persistent, spin-waiting workgroups of TWO
                                              # processes time-sliced on one device (bounded waits, abort -> multi-launch repeat)
    import torch.distributed as dist
    import mogp_emulator_amd as M
    from mogp_emulator_amd import libgpgpu
    from mogp_emulator_amd.dist import ShardedMultiOutputGP
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, T, Xs = _data(n_out)
    libgpgpu.set_fit_options(max_iter=40, ftol=1e-9, gtol=1e-6, seed=1)
    sh = ShardedMultiOutputGP(X, T, nugget="fit")
    theta0 = np.array([1.0, 1.0, 1.0, 0.0, np.log(1e-3)])
    sh.fit_GP_MAP(n_tries=1, theta0=theta0)
    mean, unc, der = sh.predict(Xs)               # the reference's default: deriv=True (MultiOutputGP_GPU.py:185-186)
    q.put((rank, (sh.lo, sh.hi), sh.get_indices_fit(), sh.get_indices_not_fit(), [None if t is None else t.copy() for t in sh.theta_hat],
           sh.logpost.copy(), sh.nuggets.copy(), mean, unc, _aborts(), der))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("chol", ["left", None])
@pytest.mark.parametrize("n_out", [5, 2])
def test_two_ranks_on_one_gpu_match_the_unsharded_model(n_out, chol):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_out, q, chol)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs: p.join(timeout=60)
    # the unsharded model, same schedule, same optimiser settings, in a third process-local engine
    if chol:
        os.environ["MOGP_CHOL"] = chol
    else:
        os.environ.pop("MOGP_CHOL", None)
    aborts = [r[9] for r in res]
    assert all(isinstance(a, int) and a >= 0 for a in aborts)            # the counter is reported by every rank
    # the one-launch Cholesky is bit-identical whatever the batch size; an aborted launch is repeated by a multi-launch
    # schedule that agrees to rounding, which the optimiser may amplify: same bar with headroom when an abort occurred
    loose = 100. if (not chol and sum(aborts) > 0) else 1.
    import mogp_emulator_amd as M
    from mogp_emulator_amd import libgpgpu
    X, T, Xs = _data(n_out)
    libgpgpu.set_fit_options(max_iter=40, ftol=1e-9, gtol=1e-6, seed=1)
    full = M.MultiOutputGP_GPU(X, T, nugget="fit")
    full = M.fit_GP_MAP(full, n_tries=1, theta0=np.array([1.0, 1.0, 1.0, 0.0, np.log(1e-3)]))
    rec = full.fit_record()
    fmean, func, fder = full.predict(Xs)
    assert res[0][1][0] == 0 and res[0][1][1] == res[1][1][0] and res[1][1][1] == n_out
    for r in res:                                   # every rank holds the records of ALL emulators
        _, _, fit_idx, notfit_idx, theta_hat, logpost, nuggets, mean, unc, _, der = r
        assert fit_idx == full.get_indices_fit() and notfit_idx == full.get_indices_not_fit()
        for k in range(n_out):
            if rec["fit_ok"][k]:
                np.testing.assert_allclose(theta_hat[k], rec["theta"][k], rtol=1e-9 * loose, atol=1e-9 * loose)
                np.testing.assert_allclose(logpost[k], rec["logpost"][k], rtol=1e-10 * loose)
                np.testing.assert_allclose(nuggets[k], rec["nugget"][k], rtol=1e-9 * loose)
            else:
                assert theta_hat[k] is None and np.isnan(logpost[k])
        np.testing.assert_allclose(mean, fmean, rtol=1e-8 * loose, atol=1e-10 * loose)
        np.testing.assert_allclose(unc, func, rtol=1e-6 * loose, atol=1e-9 * loose)
        assert der.shape == (n_out, Xs.shape[0], X.shape[1])
        np.testing.assert_allclose(der, fder, rtol=1e-7 * loose, atol=1e-8 * loose * np.abs(fder).max())
    # both ranks return the same bits
    assert np.array_equal(res[0][7], res[1][7]) and np.array_equal(res[0][8], res[1][8]) and np.array_equal(res[0][10], res[1][10])


def _nccl_world1_worker(port, q):
    """ONE rank, nccl backend, on the one GPU: RCCL initialisation + all_gather_into_tensor on device memory with the real
    payload shapes, and the device-resident predict path of ShardedMultiOutputGP (predict_dev -> gather -> one D2H copy) against
    the plain model: means / variances / input derivatives, with and without a mean function, with an emulator that is not fit."""
    try:
        import torch
        import torch.distributed as dist
        import mogp_emulator_amd as M
        from mogp_emulator_amd.dist import ShardedMultiOutputGP, gather_rows, REC_WIDTH
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        assert dist.get_backend() == "nccl"
        out = {}
        for name, shape in (("fit_records", (64, REC_WIDTH)), ("predictions", (64, 2 * 10000 + 1))):
            payload = torch.randn(shape, dtype=torch.float64, device=dev)
            got = gather_rows(payload, shape[0])                     # device defaults to the rank's GPU with nccl
            torch.cuda.synchronize()
            out[name] = bool(got.is_cuda and torch.equal(got, payload))
        X, T, Xs = _data(5)
        theta = np.tile(np.array([1.0, 1.0, 1.0, 0.0, np.log(1e-3)]), (5, 1))
        sh = ShardedMultiOutputGP(X, T, nugget="fit")
        assert sh._device_path()
        sh.fit(theta)
        mean, unc, der = sh.predict(Xs)
        full = M.MultiOutputGP_GPU(X, T, nugget="fit")
        full.fit(theta)
        fmean, func, fder = full.predict(Xs)
        out["predict"] = bool(np.array_equal(mean, fmean) and np.allclose(unc, func, rtol=0, atol=1e-15) and np.array_equal(der, fder))
        out["fit"] = sh.get_indices_fit() == [0, 1, 2, 3, 4] and bool(np.allclose(sh.nuggets, 1e-3))
        m2, u2, d2 = sh.predict(Xs, deriv=False, include_nugget=False)
        f2m, f2u, _ = full.predict(Xs, deriv=False, include_nugget=False)
        out["predict_no_nugget"] = bool(np.array_equal(m2, f2m) and np.allclose(u2, f2u, rtol=0, atol=1e-15) and not d2.any())
        m3, u3, d3 = sh.predict(Xs, unc=False)
        out["predict_no_unc"] = bool(np.array_equal(m3, fmean) and not u3.any() and np.array_equal(d3, fder))
        try:
            sh.predict(Xs, full_cov=True)
            out["full_cov_refused"] = False
        except NotImplementedError:
            out["full_cov_refused"] = True
        # a mean function (in theta, or integrated out analytically): the mean terms are added on the device, the gather still runs
        # on device buffers (densegp_gpu.hpp:300-408, GaussianProcess.py:885-920)
        for tag, kw, n_mean in (("theta_mean", {"mean": "c+c*x[0]+c*x[1]^2"}, 3), ("const_mean", {"mean": "c"}, 1),
                                ("analytic_mean", {"mean": "c+c*x[0]", "analytic_mean": True}, 0)):
            shm = ShardedMultiOutputGP(X, T, nugget=1e-3, **kw)
            assert shm._device_path()
            fm = M.MultiOutputGP_GPU(X, T, nugget=1e-3, **kw)
            th = np.tile(np.concatenate([np.linspace(0.3, -0.2, n_mean), [1.0, 1.0, 1.0, 0.0]]), (5, 1))
            shm.fit(th)
            fm.fit(th)
            mm, mu, md = shm.predict(Xs)
            rm, ru, rd = fm.predict(Xs)
            out["predict_" + tag] = bool(np.array_equal(mm, rm) and np.allclose(mu, ru, rtol=0, atol=1e-15) and np.array_equal(md, rd))
        # an emulator that is not fit: ValueError unless allow_not_fit, then NaN rows (MultiOutputGP_GPU.py:267-296)
        shn = ShardedMultiOutputGP(X, T, nugget=1e-3)
        shn.fit(theta[:, :4])
        shn.local.reset_fit_status()
        shn.local.fit_emulator(0, theta[0, :4]); shn.local.fit_emulator(3, theta[3, :4])
        shn._gather_fit_records()
        out["not_fit_indices"] = shn.get_indices_not_fit() == [1, 2, 4]
        try:
            shn.predict(Xs)
            out["not_fit_raises"] = False
        except ValueError:
            out["not_fit_raises"] = True
        nm, nu, nd = shn.predict(Xs, allow_not_fit=True)
        fn = M.MultiOutputGP_GPU(X, T, nugget=1e-3)
        fn.fit(theta[:, :4])
        rm, ru, rd = fn.predict(Xs)
        out["not_fit_nan_rows"] = bool(np.isnan(nm[[1, 2, 4]]).all() and np.isnan(nu[[1, 2, 4]]).all() and np.isnan(nd[[1, 2, 4]]).all()
                                       and np.array_equal(nm[[0, 3]], rm[[0, 3]]) and np.allclose(nu[[0, 3]], ru[[0, 3]], rtol=0, atol=1e-15)
                                       and np.array_equal(nd[[0, 3]], rd[[0, 3]]))
        dist.destroy_process_group()
        q.put(out)
    except Exception as exc:                                          # noqa: BLE001
        import traceback
        q.put({"error": traceback.format_exc()[-3000:]})


def test_rccl_world1_gathers_the_real_payloads_on_device():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=60)
    assert "error" not in res, res.get("error")
    assert all(res.values()), res


def _bench_line(gpus, extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MOGP_CHOL")}
    env["MOGP_BENCH_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0"] + extra,
                         env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher around it) must start two ranks itself (--gpus was parsed and
    ignored).  gloo + both ranks on device 0, as the one GPU of this box allows; the command shape is the driver's.  Round 6: an
    N > 1 line carries `cpu_baseline` (rank 0, after the timed region) and, nested in `roofline`, the headline kernel's fraction with
    the kernel times taken as maxima over the ranks; `traffic` is null and says why."""
    out = _bench_line(2, [])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["collective_backend"] == "gloo"
    assert out["config"]["outputs_per_gpu"] == 32 and out["config"]["outputs_total"] == 64
    assert np.isfinite(out["value"]) and out["value"] > 0 and np.isfinite(out["predict_pts_per_s"])
    assert np.isfinite(out["logpost_checksum"])
    rf = out["roofline"]
    assert rf["traffic"] is None and "N > 1" in rf["traffic_source"]
    assert rf["headline_kernel"]["kernel"] == "mchol" and 0. < rf["headline_kernel"]["frac"] < 1.
    assert rf["kernels"]["mchol"]["avg_ms"] > 0. and "max over the 2 ranks" in out["kernels"]["mchol"]["note"]
    assert out["kernels"]["mchol"]["ms_total"] >= out["kernels"]["mchol"]["ms_total_this_rank"]
    assert rf["fit_phase"]["peak"] == 2 * 78.6
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["parity_in_bench"]["passed"] is True


def test_bench_eight_ranks_control_flow_on_one_gpu():
    """The driver's N = 8 line (8 emulators per rank: the chain-bound regime of the one-launch Cholesky), all eight ranks on the one
    device of this box over gloo: shard bounds, the padded gather, the max-over-ranks reductions and the line's shape."""
    out = _bench_line(8, ["--no-cpu-baseline", "--m", "2000"])
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8
    assert out["config"]["outputs_per_gpu"] == 8 and out["config"]["outputs_total"] == 64 and out["scaling"] == "strong"
    assert np.isfinite(out["value"]) and out["value"] > 0 and np.isfinite(out["logpost_checksum"])
    assert out["roofline"]["headline_kernel"]["flops_per_launch"] == 8 * 2000. ** 3 / 3.
    assert "cpu_baseline" not in out
