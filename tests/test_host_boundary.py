"""CPU-only tests: the C-ABI library loads and exports every symbol include/mogp_hip.h declares,
and the host-side mirror of the reference's GPU-facing interface behaves like the reference's
(names, argument coercion, error types).  No compute calls are made without a GPU."""
import os
import pickle
import re

import numpy as np
import pytest
from numpy.testing import assert_allclose

import mogp_emulator_amd as M
from mogp_emulator_amd import LibGPGPU, _capi, libgpgpu
from mogp_emulator_amd.GaussianProcessGPU import (PredictResult, create_prior_params, interpret_nugget,
                                                   ndarray_coerce_type_and_flags, parse_meanfunc_formula)
from mogp_emulator_amd.Priors import GammaPrior, GPPriors, InvGammaPrior, LogNormalPrior, WeakPrior, input_spacing
from conftest import ROOT, load_golden


def test_library_loads_and_exports_every_declared_symbol():
    lib = _capi.load()
    header = open(os.path.join(ROOT, "include", "mogp_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mogp_[a-zA-Z0-9_]+)\s*\(", header))
    assert len(declared) > 70
    for name in declared:
        assert hasattr(lib, name), "libmogp_hip.so does not export " + name
        assert name in _capi.SIGNATURES, "no ctypes prototype for " + name
    assert set(_capi.SIGNATURES) == declared
    assert b"gfx950" in lib.mogp_version()


def test_no_gpu_here_means_loud_failure():
    if LibGPGPU.gpu_usable():
        pytest.skip("a GPU is visible")
    assert LibGPGPU.HAVE_LIBGPGPU
    with pytest.raises(RuntimeError):
        M.GaussianProcessGPU(np.zeros((3, 2)), np.zeros(3))
    with pytest.raises(RuntimeError):
        M.MultiOutputGP_GPU(np.zeros((3, 2)), np.zeros((2, 3)))


def test_module_surface_matches_reference_bindings():
    # exported names of bindings.cu:13-621 that the Python layer uses
    for name in ["have_compatible_device", "fit_GP_MAP", "kernel_type", "nugget_type", "prior_type", "ZeroMeanFunc",
                 "FixedMeanFunc", "ConstMeanFunc", "PolyMeanFunc", "GPParameters", "DenseGP_GPU", "MultiOutputGP_GPU",
                 "SquaredExponentialKernel", "Matern52Kernel", "WeakPrior", "InvGammaPrior", "GammaPrior", "CovTransform",
                 "CorrTransform", "GPPriors"]:
        assert hasattr(LibGPGPU, name), name
    for meth in ["n", "D", "n_corr", "inputs", "targets", "n_params", "theta_fit_status", "reset_theta_fit_status", "get_theta",
                 "get_gppriors", "create_gppriors", "predict", "predict_variance", "predict_batch", "predict_variance_batch",
                 "predict_deriv", "fit", "get_K", "get_invQ", "get_invQt", "get_logpost", "get_nugget_size", "set_nugget_size",
                 "get_nugget_type", "set_nugget_type", "get_kernel_type", "get_meanfunc", "get_cholesky_lower", "logpost_deriv"]:
        assert hasattr(libgpgpu.DenseGP_GPU, meth), meth
    for meth in ["inputs", "targets", "targets_at_index", "emulator", "n", "D", "n_emulators", "n_data_params", "n_corr_params",
                 "get_nugget_type", "get_nugget_size", "get_fitted_indices", "get_unfitted_indices", "create_priors_for_emulator",
                 "reset_fit_status", "fit_emulator", "fit", "predict", "predict_batch", "predict_variance_batch", "predict_deriv"]:
        assert hasattr(libgpgpu.MultiOutputGP_GPU, meth), meth


def test_enums_behave_like_pybind_enums():
    nt = LibGPGPU.nugget_type
    assert str(nt.adaptive) == "nugget_type.adaptive" and str(nt.fit).split(".")[1] == "fit"
    assert nt(0) == nt.adaptive and nt(1) == nt.fit and nt(2) == nt.fixed          # types.hpp:29
    assert int(LibGPGPU.kernel_type.SquaredExponential) == 0 and int(LibGPGPU.kernel_type.Matern52) == 1
    pt = LibGPGPU.prior_type
    assert [int(pt.InvGamma), int(pt.Gamma), int(pt.LogNormal), int(pt.Weak)] == [0, 1, 2, 3]
    with pytest.raises(ValueError):
        nt(7)


def test_mean_functions_and_error_messages():
    # tests/test_GPUMeanFunction.py:87-109 of the reference
    x = np.array([[1., 2.], [3., 4.]])
    assert_allclose(LibGPGPU.ZeroMeanFunc().mean_f(x, np.zeros(0)), [0., 0.])
    assert_allclose(LibGPGPU.FixedMeanFunc(2.5).mean_f(x, np.zeros(0)), [2.5, 2.5])
    assert_allclose(LibGPGPU.ConstMeanFunc().mean_f(x, np.array([4.4])), [4.4, 4.4])
    poly = LibGPGPU.PolyMeanFunc([[0, 1], [1, 2]])
    assert poly.get_n_params() == 3
    assert_allclose(poly.mean_f(x, np.array([1., 2., 3.])), [1 + 2 * 1 + 3 * 4, 1 + 2 * 3 + 3 * 16])
    assert_allclose(poly.mean_deriv(x, np.array([1., 2., 3.])), [[1., 1.], [1., 3.], [4., 16.]])
    assert_allclose(poly.mean_inputderiv(x, np.array([1., 2., 3.])), [[2., 2.], [3 * 2 * 2., 3 * 2 * 4.]])
    with pytest.raises(RuntimeError, match="Expected params list of length 3"):
        poly.mean_f(x, np.array([1.]))
    with pytest.raises(RuntimeError, match="Expected params list of length 1"):
        LibGPGPU.ConstMeanFunc().mean_f(x, np.zeros(0))
    with pytest.raises(RuntimeError, match="Dimension index must be less than"):
        LibGPGPU.PolyMeanFunc([[5, 1]]).mean_f(x, np.array([1., 2.]))


def test_parse_meanfunc_formula():
    assert isinstance(parse_meanfunc_formula("c"), LibGPGPU.ConstMeanFunc)
    assert isinstance(parse_meanfunc_formula("3.5"), LibGPGPU.FixedMeanFunc)
    p = parse_meanfunc_formula("c+c*x[0]+c*x[1]^2")
    assert isinstance(p, LibGPGPU.PolyMeanFunc) and p.get_n_params() == 3
    assert_allclose(p.mean_f(np.array([[2., 3.]]), np.array([1., 1., 1.])), [1 + 2 + 9])
    assert parse_meanfunc_formula("c+c*x[0]*x[0]").mean_f(np.array([[3.]]), np.array([0., 1.]))[0] == 9.
    with pytest.raises(NotImplementedError):
        parse_meanfunc_formula("c*x[0]*x[1]")
    assert parse_meanfunc_formula("hello") is None


def test_gpparameters_host_object():
    p = LibGPGPU.GPParameters(0, 3, LibGPGPU.nugget_type.fit)
    assert p.get_n_data() == 5 and not p.data_has_been_set()
    p.set_data(np.array([0., 2., -2., 1., np.log(1e-6)]))
    assert p.data_has_been_set()
    assert_allclose(p.get_corr(), np.exp(-0.5 * np.array([0., 2., -2.])))       # CorrTransform, GPParams.py:35-45
    assert_allclose(p.get_cov(), np.e)
    assert_allclose(p.get_nugget_size(), 1e-6)
    assert p.test_same_shape(np.zeros(5)) and not p.test_same_shape(np.zeros(4))
    with pytest.raises(RuntimeError):
        p.set_data(np.zeros(3))
    p.unset_data()
    assert not p.data_has_been_set() and p.get_cov() == 0.
    q = LibGPGPU.GPParameters(0, 3, LibGPGPU.nugget_type.fixed, 1e-3)
    assert q.get_n_data() == 4 and q.get_nugget_size() == 1e-3


def test_transforms():
    c = LibGPGPU.CorrTransform()
    assert_allclose(c.raw_to_scaled(2.), np.exp(-1.)); assert_allclose(c.scaled_to_raw(np.exp(-1.)), 2.)
    assert_allclose(c.dscaled_draw(3.), -1.5); assert_allclose(c.d2scaled_draw2(2.), 0.5)   # CPU sign, GPParams.py:69-80
    v = LibGPGPU.CovTransform()
    assert_allclose(v.raw_to_scaled(1.), np.e); assert_allclose(v.dscaled_draw(3.), 3.)


def test_interpret_nugget_and_coercion():
    assert interpret_nugget("adaptive") == (LibGPGPU.nugget_type.adaptive, 0.)
    assert interpret_nugget("fit") == (LibGPGPU.nugget_type.fit, 0.)
    assert interpret_nugget(1e-4) == (LibGPGPU.nugget_type.fixed, 1e-4)
    assert interpret_nugget(1) == (LibGPGPU.nugget_type.fixed, 1.)
    assert interpret_nugget("pivot") == (LibGPGPU.nugget_type.pivot, 0.)     # CPU-class mode, GPParams.py:185-186
    with pytest.raises(ValueError):
        interpret_nugget("nonsense")
    with pytest.raises(ValueError):
        interpret_nugget(-1.)
    with pytest.raises(TypeError):
        interpret_nugget([1, 2])
    a = ndarray_coerce_type_and_flags(np.arange(6).reshape(2, 3)[:, ::2])
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"]


def test_priors_match_reference_values():
    g = load_golden("priors.npz")
    for nm, cls in (("invgamma", InvGammaPrior), ("gamma", GammaPrior), ("lognormal", LogNormalPrior)):
        p = cls(2., 2.)
        assert_allclose([p.logp(x) for x in g["x"]], g[nm + "_2_2_logp"], rtol=1e-13)
        assert_allclose([p.dlogpdx(x) for x in g["x"]], g[nm + "_2_2_dlogpdx"], rtol=1e-13)
    assert WeakPrior().logp(1.) == 0.


def test_default_priors_match_reference():
    g = load_golden("priors.npz")
    X = np.random.default_rng(int(g["default_X_seed"])).uniform(0, 1, (2000, 10))
    lo_hi = np.array([input_spacing(c) for c in X.T])
    assert_allclose(lo_hi[:, 0], g["default_min_spacing"], rtol=1e-14)
    assert_allclose(lo_hi[:, 1], g["default_max_spacing"], rtol=1e-14)
    pri = GPPriors.default_priors(X, 10, "fit")
    assert_allclose([p.shape for p in pri.corr], g["default_corr_shape"], rtol=1e-9)
    assert_allclose([p.scale for p in pri.corr], g["default_corr_scale"], rtol=1e-9)
    assert_allclose([pri.nugget.shape, pri.nugget.scale], g["default_nugget"], rtol=1e-9)
    # SURVEY 8b pinning I/O
    assert_allclose(pri.corr[0].shape, 0.8408705121074469, rtol=1e-9)
    assert_allclose(pri.corr[0].scale, 0.0017112471182106296, rtol=1e-9)
    assert isinstance(pri.cov, WeakPrior) and not isinstance(pri.cov, InvGammaPrior)
    assert GPPriors.default_priors(X, 10, "adaptive").nugget is None
    # too few unique inputs -> weak (Priors.py:746-779)
    flat = GPPriors.default_priors(np.ones((5, 2)), 2, "fixed")
    assert all(type(p) is WeakPrior for p in flat.corr)


def test_create_prior_params_layout():
    X = np.random.default_rng(1).uniform(0, 1, (50, 3))
    n_corr, corr, cov, nug = create_prior_params(inputs=X, n_corr=3, nugget_type="fit")
    assert n_corr == 3 and len(corr) == 3
    assert corr[0][0] == LibGPGPU.prior_type.InvGamma and len(corr[0][1]) == 2
    assert cov == (LibGPGPU.prior_type.Weak, [0., 0.])
    assert nug[0] == LibGPGPU.prior_type.InvGamma
    n_corr, corr, cov, nug = create_prior_params(newpriors=GPPriors(n_corr=2, nugget_type="fixed", cov=GammaPrior(2., 3.)))
    assert [c[0] for c in corr] == [LibGPGPU.prior_type.Weak] * 2 and cov == (LibGPGPU.prior_type.Gamma, [2., 3.])
    n_corr, corr, cov, nug = create_prior_params(newpriors=dict(n_corr=1, nugget_type="fit", nugget=LogNormalPrior(1., 2.)))
    assert nug == (LibGPGPU.prior_type.LogNormal, [1., 2.])
    with pytest.raises(TypeError):
        create_prior_params(foo=1)
    with pytest.raises(TypeError):
        create_prior_params(newpriors=dict(bogus=3))


def test_gppriors_validation():
    with pytest.raises(ValueError):
        GPPriors()
    with pytest.raises(AssertionError):
        GPPriors(n_corr=1, nugget_type="blah")
    with pytest.raises(TypeError):
        GPPriors(corr=[1., 2.])
    with pytest.raises(TypeError):
        GPPriors(n_corr=1, cov=3.)
    p = GPPriors(n_corr=2, nugget_type="adaptive", nugget=InvGammaPrior(1., 1.))
    assert p.nugget is None and p.n_corr == 2


def test_predict_result_container():
    r = PredictResult(mean=np.ones(2), unc=None, deriv=np.zeros((2, 1)))
    mean, unc, deriv = r
    assert unc is None and r[0] is r.mean and r["deriv"] is r[2]
    with pytest.raises(KeyError):
        r[3]
    with pytest.raises(AttributeError):
        r.nothing
    assert pickle.loads(pickle.dumps(dict(r)))["unc"] is None


def test_fit_gp_map_argument_checks():
    from mogp_emulator_amd.fitting import fit_GP_MAP, _check_common
    with pytest.raises(TypeError):
        fit_GP_MAP()
    with pytest.raises(NotImplementedError):
        _check_common(3, "Nelder-Mead")
    with pytest.raises(AssertionError):
        _check_common(0, "L-BFGS-B")


def test_constructible_native_gppriors_match_the_reference():
    """GPPriors(n_corr, nugget_type) of the native module (mogp_gpu/src/bindings.cu:528-556): logp / dlogpdtheta /
    d2logpdtheta2 against the reference's values (tests/golden/make_golden_priors2.py), both ways of filling it."""
    from mogp_emulator_amd import libgpgpu as L
    g = load_golden("gppriors_native.npz")
    for nm, cls in (("invgamma", L.InvGammaPrior), ("gamma", L.GammaPrior), ("lognormal", L.LogNormalPrior)):
        p = cls(2.5, 0.7)
        assert_allclose([p.d2logpdx2(x) for x in g["x"]], g[nm + "_d2logpdx2"], rtol=1e-13)
    for k in range(int(g["n_cases"])):
        nt = str(g["c%d_nugget_type" % k])
        corr, cov, nug, th = g["c%d_corr" % k], g["c%d_cov" % k], g["c%d_nug" % k], g["c%d_theta" % k]
        ntype = getattr(L.nugget_type, nt)
        for way in ("create", "set"):
            pri = L.GPPriors(len(corr), ntype)
            if way == "create":
                pri.create_corr_priors([(L.prior_type(int(c[0])), [c[1], c[2]]) for c in corr])
                pri.create_cov_prior((L.prior_type(int(cov[0])), [cov[1], cov[2]]))
                pri.set_nugget((L.prior_type(int(nug[0])), [nug[1], nug[2]]))
            else:
                pri.set_corr([L.GPPriors.make_prior(int(c[0]), [c[1], c[2]]) for c in corr])
                pri.set_cov(L.GPPriors.make_prior(int(cov[0]), [cov[1], cov[2]]))
                pri.set_nugget(L.GPPriors.make_prior(int(nug[0]), [nug[1], nug[2]]))
            assert len(pri.get_corr()) == len(corr) and pri.get_nugget_type() == ntype
            assert (pri.get_nugget() is not None) == (nt == "fit")
            theta = L.GPParameters(0, len(corr), ntype, 1e-6 if nt == "fixed" else 0.)
            theta.set_data(th)
            assert_allclose(pri.get_logp(theta), float(g["c%d_logp" % k]), rtol=1e-12)
            assert_allclose(pri.get_dlogpdtheta(theta), g["c%d_dlogp" % k], rtol=1e-12, atol=1e-14)
            assert_allclose(pri.get_d2logpdtheta2(theta), g["c%d_d2logp" % k], rtol=1e-12, atol=1e-14)
            smp = pri.sample()
            assert len(smp) == len(th) and np.all(np.isfinite(smp))
    # defaults and errors of the container
    pri = L.GPPriors(3, L.nugget_type.fit)
    pri.set_corr(); pri.set_cov(); pri.set_nugget()
    assert all(type(p) is L.WeakPrior for p in pri.get_corr()) and type(pri.get_cov()) is L.WeakPrior and type(pri.get_nugget()) is L.WeakPrior
    theta = L.GPParameters(0, 3, L.nugget_type.fit)
    theta.set_data(np.array([0.1, 0.2, 0.3, 0.4, -5.]))
    assert pri.get_logp(theta) == 0. and np.all(pri.get_d2logpdtheta2(theta) == 0.)
    with pytest.raises(RuntimeError):
        pri.get_logp(L.GPParameters(0, 2, L.nugget_type.fit))          # no data / wrong shape
    assert type(L.GPPriors.make_prior(L.prior_type.InvGamma, [1.0])) is L.WeakPrior      # wrong parameter count -> weak
    fixed = L.GPPriors(1, L.nugget_type.fixed)
    fixed.set_nugget((L.prior_type.Gamma, [2., 2.]))
    assert fixed.get_nugget() is None                                   # only a fitted nugget carries a prior


def test_native_meanpriors_prior_dists_and_sample():
    from mogp_emulator_amd import libgpgpu as L
    mp = L.MeanPriors([1., 2.], [[2., 0.], [0., 3.]])
    s = mp.sample(L.CorrTransform())
    assert len(s) == 2 and all(-2.5 <= v <= 2.5 for v in s)             # weak: raw draw from U(-2.5, 2.5)
    mp.set_prior_dists([L.prior_type.Gamma, L.prior_type.LogNormal], [[2., 1.], [0.5, 1.]])
    np.random.seed(3)
    s = mp.sample(L.CovTransform())
    assert len(s) == 2 and np.all(np.isfinite(s))
    with pytest.raises(RuntimeError, match="must equal number of meanfunc parameters"):
        mp.set_prior_dists([L.prior_type.Gamma], [[2., 1.]])
    mp.set_prior_dists()
    assert all(type(d) is L.WeakPrior for d in mp._dists)
    pri = L.GPPriors(1, L.nugget_type.adaptive)
    pri.set_corr(); pri.set_cov(); pri.set_mean(mp)
    assert len(pri.sample()) == 2 + 1 + 1                               # mean parameters first (gppriors.hpp:458-471)


@pytest.mark.parametrize("ahead", [False, True], ids=["in-order", "band-ahead"])
@pytest.mark.parametrize("n", [1, 100, 128, 129, 300, 640, 2000, 5000, 16000])
def test_one_launch_cholesky_task_order_is_topological(n, ahead):
    """The forward-progress argument of the one-launch Cholesky (csrc/kernels_mchol.hip) rests on ONE property of its task table:
    every task only depends on tasks with a smaller number.  Replay the dependency rules of the kernel against the tables the
    library builds (host-only entry points, no device needed):
      D(c)     needs G(0, c), G(1, c), G(2, c) (c >= 2) and the row tiles 2c, 2c+1 of column c-1 (c >= 1)
      G(s, c)  (lower 64 x 64 tile (ti, tj) = (0,0), (1,0), (1,1) of the diagonal block) needs the row tiles 2c+ti, 2c+tj of every column k <= c-2
      T(r, c)  needs D(c) and the row tiles r, 2c, 2c+1 of every column k <= c-1
    and check that every tile of the lower block triangle is produced exactly once.
    The band-ahead table (round 5) has ONE bounded exception, and only for entries that carry bit 29: such a T task may stand in front of
    its D(c) -- at most 6 places, with nothing but flagged entries in between (the launcher's condition for using the table, more
    workgroups per queue than 7 tickets per emulator, and the kernel's "a flagged task draws no ticket while it waits" rest on this)."""
    import ctypes
    lib = _capi.load()
    fn = lib.mogp_mchol_task_table_ahead if ahead else lib.mogp_mchol_task_table
    cnt = fn(n + 1, None, 0)
    buf = (ctypes.c_int * cnt)()
    assert fn(n + 1, buf, cnt) == cnt
    NP = (n + 1 + 127) // 128 * 128
    K, K2 = NP // 128, NP // 64
    pos = {}
    flagged = set()
    tile = {}                                   # (row tile, column) -> position of the task that produces it
    for p, w in enumerate(buf):
        w &= 0xffffffff
        t, c, r = (w >> 30) & 3, (w >> 15) & 0x3fff, w & 0x7fff
        if (w >> 29) & 1:
            assert ahead and t == 2
            flagged.add(p)
        key = (t, c, r if t else 0)
        assert key not in pos, "task listed twice: %r" % (key,)
        pos[key] = p
        assert t != 3
        if t == 2:
            assert (r, c) not in tile, "tile produced twice: %r" % ((r, c),)
            tile[(r, c)] = p
    assert sorted(c for (t, c, r) in pos if t == 0) == list(range(K))
    assert sorted(tile) == sorted((r, c) for c in range(K) for r in range(2 * c + 2, K2))
    assert sorted((r, c) for (t, c, r) in pos if t == 1) == sorted((sub, c) for c in range(2, K) for sub in range(3))
    for (t, c, r), p in pos.items():
        need_tiles, deps = [], []
        if t == 0:
            if c >= 2:
                deps += [(1, c, 0), (1, c, 1), (1, c, 2)]
            if c >= 1:
                need_tiles += [(2 * c, c - 1), (2 * c + 1, c - 1)]
        elif t == 1:
            ti, tj = (1 if r > 0 else 0), (1 if r > 1 else 0)
            for k in range(c - 1):
                need_tiles += [(rr, k) for rr in {2 * c + ti, 2 * c + tj}]
        else:
            rows = {r, 2 * c, 2 * c + 1}
            for k in range(c):
                need_tiles += [(rr, k) for rr in rows]
            deps.append((0, c, 0))
        for d in deps:
            if t == 2 and p in flagged and d in pos and pos[d] > p:
                assert pos[d] - p <= 6 and all(q in flagged for q in range(p + 1, pos[d])), "flagged task %r at %d, its D at %d" % ((t, c, r), p, pos[d])
                continue
            assert d in pos and pos[d] < p, "task %r (position %d) depends on %r (position %s)" % ((t, c, r), p, d, pos.get(d))
        for d in need_tiles:
            assert d in tile and tile[d] < p, "task %r (position %d) needs tile %r (position %s)" % ((t, c, r), p, d, tile.get(d))


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    """bench.py --gpus N starts N ranks itself when no launcher set WORLD_SIZE, and refuses a launcher whose world size differs
    (--gpus used to be parsed and ignored).  Decided before torch is imported: runs without a GPU."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=60)
    assert res.returncode != 0 and "WORLD_SIZE=3" in res.stderr


def test_one_hip_runtime_per_process_after_loading_the_library():
    """_capi.load() must leave ONE libamdhip64 in the process -- the copy bundled with an installed PyTorch-ROCm wheel when there is one,
    so that torch (imported later, or earlier) and the library share a runtime (round 4: with two runtimes torch.cuda found no device
    after a fit on some MI355X hosts).  Checked in a fresh interpreter, in both import orders; needs no GPU."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, os
sys.path.insert(0, %r)
order = sys.argv[1]
for w in order:
    if w == "t":
        import torch
    else:
        from mogp_emulator_amd import _capi
        _capi.load()
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
print("HIPLIBS", len(libs), libs)
""" % root
    for order in ("l", "lt", "tl"):
        out = subprocess.run([sys.executable, "-c", code, order], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("HIPLIBS")][0]
        assert line.split()[1] == "1", (order, line)


def test_lean_exp_matches_long_double(tmp_path):
    """The covariance kernels' 14-instruction exponential (csrc/exp_dev.h; replaces ocml's exp, Kernel.py:772-791, 861-882 use np.exp):
    the same source text compiled for the host, against long double expl.  <= 1.1 ulp on normal results, exp(-x/2) form bit-identical
    to the exp(-x) form at x/2, libm's behaviour at 0 / underflow / inf / NaN."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "exp_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "mogp_emulator_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "exp_check.cpp"), "-o", exe])
    full, half, mismatch, ok = subprocess.check_output([exe, "4000000"]).decode().split()
    assert float(full) <= 1.1 and float(half) <= 1.1, (full, half)
    assert int(mismatch) == 0 and int(ok) == 1
    # the table in the header is what tools/gen_exp_table.py would write
    tab = open(os.path.join(ROOT, "mogp_emulator_amd", "csrc", "exp_tab.h")).read()
    vals = [float.fromhex(h) for h in re.findall(r"0x1\.[0-9a-f]+p\+0", tab)]
    assert len(vals) == 256
    assert_allclose(vals, 2.0 ** (np.arange(256) / 256.0), rtol=3e-16)


def test_hip_runtime_preload_only_for_matching_soname(tmp_path, monkeypatch):
    """the loader puts a PyTorch wheel's bundled HIP runtime in front of libmogp_hip.so only when that copy's DT_SONAME is
    the libamdhip64.so.N this library was linked against; another ROCm major is left alone."""
    import importlib.util
    import shutil
    import subprocess
    import types
    soname, needed = _capi.elf_dynamic(_capi.LIB_PATH)
    hip_needed = [n for n in needed if n.startswith("libamdhip64.so")]
    assert len(hip_needed) == 1 and re.fullmatch(r"libamdhip64\.so\.\d+", hip_needed[0])
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    # a fake torch package whose lib/ holds a libamdhip64.so with a chosen SONAME
    def fake_torch(tag, so):
        root = tmp_path / tag / "torch"
        (root / "lib").mkdir(parents=True)
        (root / "__init__.py").write_text("")
        src = tmp_path / (tag + ".c")
        src.write_text("int mogp_fake_runtime(void) { return 1; }\n")
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-Wl,-soname," + so, str(src), "-o", str(root / "lib" / "libamdhip64.so")])
        return types.SimpleNamespace(origin=str(root / "__init__.py"))
    same, other = fake_torch("same", hip_needed[0]), fake_torch("other", "libamdhip64.so.6")
    assert _capi.elf_dynamic(os.path.join(os.path.dirname(other.origin), "lib", "libamdhip64.so"))[0] == "libamdhip64.so.6"
    monkeypatch.setattr(importlib.util, "find_spec", lambda name: same)
    assert _capi._bundled_runtime_to_preload(_capi.LIB_PATH) == [os.path.join(os.path.dirname(same.origin), "lib", "libamdhip64.so")]
    monkeypatch.setattr(importlib.util, "find_spec", lambda name: other)
    assert _capi._bundled_runtime_to_preload(_capi.LIB_PATH) == []
    monkeypatch.setattr(importlib.util, "find_spec", lambda name: None)
    assert _capi._bundled_runtime_to_preload(_capi.LIB_PATH) == []
    # a truncated / malformed bundled libamdhip64.so degrades to "nothing to preload" instead of failing the import
    good = open(os.path.join(os.path.dirname(same.origin), "lib", "libamdhip64.so"), "rb").read()
    for tag, blob in (("cut_phdr", good[:0x60]), ("cut_ident", good[:40]), ("small_phent", good[:0x36] + b"\x10\x00" + good[0x38:]), ("junk", b"\x7fELF\x02\x01" + b"\xff" * 80)):
        root = tmp_path / tag / "torch"
        (root / "lib").mkdir(parents=True)
        (root / "__init__.py").write_text("")
        (root / "lib" / "libamdhip64.so").write_bytes(blob)
        broken = types.SimpleNamespace(origin=str(root / "__init__.py"))
        monkeypatch.setattr(importlib.util, "find_spec", lambda name, b=broken: b)
        assert _capi._bundled_runtime_to_preload(_capi.LIB_PATH) == [], tag
    monkeypatch.setenv("MOGP_HIP_RUNTIME", "system")
    monkeypatch.setattr(importlib.util, "find_spec", lambda name: same)
    assert _capi._one_hip_runtime_per_process() == []
