"""fit_GP_MAP(n_tries starts) wall time, evaluation counts and the tagged device time of the fit kernels.
    env: B (64), N (2000), D (10), TRIES (15), MAXITER (10), NUGGET (adaptive)"""
import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd import libgpgpu, _capi
from bench import synth, counter
B, n, d, tries, mi = (int(os.environ.get(k, v)) for k, v in (("B", 64), ("N", 2000), ("D", 10), ("TRIES", 15), ("MAXITER", 10)))
nug = os.environ.get("NUGGET", "adaptive")
try:
    nug = float(nug)
except ValueError:
    pass
lib = _capi.load()
X, T, _ = synth(2, n, d, B, 8)
libgpgpu.set_fit_options(max_iter=mi, ftol=1e-9, gtol=1e-6, seed=1)
for rep in range(2):
    gp = M.MultiOutputGP_GPU(X, T, nugget=nug)
    e0, g0 = counter("objective_evals"), counter("gradient_evals")
    r0, sr0, b0 = counter("pool_rounds"), counter("pool_slot_rounds"), counter("replica_engine_build_us")
    lib.mogp_profile_reset(); lib.mogp_profile_enable(int(os.environ.get("PROF", "0")))
    t0 = time.perf_counter()
    libgpgpu.fit_GP_MAP(gp._mogp_gpu, tries)
    dt = time.perf_counter() - t0
    lib.mogp_profile_enable(0)
    ev, gv = counter("objective_evals") - e0, counter("gradient_evals") - g0
    print("  runs %d, accepted steps %d, trial points shortened %d / lengthened %d" % tuple(counter(k) for k in ("lbfgs_runs", "lbfgs_iterations", "linesearch_shortened", "linesearch_lengthened")))
    rounds = counter("pool_rounds") - r0
    print("  pool: %d rounds, mean batch %.1f slots; replica engine built in %.1f ms" % (rounds, (counter("pool_slot_rounds") - sr0) / max(rounds, 1), (counter("replica_engine_build_us") - b0) / 1e3))
    print("B=%d n=%d tries=%d max_iter=%d: %.3f s, %d obj / %d grad evals, %.1f TF, %.3f ms per eval" % (B, n, tries, mi, dt, ev, gv, (gv * 2.0 / 3.0 + ev / 3.0) * float(n) ** 3 / dt * 1e-12, dt / max(ev, 1) * 1e3), flush=True)
