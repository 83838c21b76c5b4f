"""Round 6: why is fit_GP_MAP 1.5 x slower inside bench.py than alone?  Times bench.time_fit_map (64 emulators x 15 starts) in a fresh process, then again after
each ingredient of the bench process: torch imported, the main model alive (its factor / L^-1 / K^-1 / K* buffers), a predict, the shard sweep's models."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd.Priors import GPPriors
import bench
X, T, Xs = bench.synth(2, 2000, 10, 64, 10000)
def run(tag, B=64):
    b0, p0, r0, n0 = (bench.counter(k) for k in ("replica_engine_build_us", "replica_pool_us", "retarget_us", "retargets"))
    r = bench.time_fit_map(M, X, T[:B], "SquaredExponential", 1e-6, 15, 10)
    print("%-60s B=%d fit_GP_MAP %.3f s (first call %.3f)  %.1f TF  (both calls: replica engines built %.3f s, pools %.3f s, of which %d retargets %.3f s)" % (
        tag, B, r["fit_GP_MAP_s"], r["fit_GP_MAP_first_call_s"], r["fit_GP_MAP_TFLOPs"], (bench.counter("replica_engine_build_us") - b0) / 1e6,
        (bench.counter("replica_pool_us") - p0) / 1e6, bench.counter("retargets") - n0, (bench.counter("retarget_us") - r0) / 1e6), flush=True)
run("fresh process")
run("again")
import torch
torch.cuda.set_device(0)
x = torch.zeros(10, device="cuda")
run("torch imported, device set, one tensor")
theta = np.array([-2. * np.log(0.3 * np.sqrt(10))] * 10 + [0.])
gp = M.MultiOutputGP_GPU(X, T, nugget=1e-6, priors=GPPriors(n_corr=10, nugget_type="fixed"))
mo = gp._mogp_gpu
mo.eval(np.tile(theta, (64, 1)), grad=True)
run("+ main model alive after eval(grad=True)")
d_Xs = torch.from_numpy(Xs).cuda(); d_mean = torch.empty((64, 10000), dtype=torch.float64, device="cuda"); d_var = torch.empty_like(d_mean)
mo.predict_variance_batch_dev(d_Xs.data_ptr(), 10000, d_mean.data_ptr(), d_var.data_ptr())
run("+ device-resident predict done (K* chunk buffer alive)")
s = bench.time_shard(M, GPPriors, 2, 2000, 10, 8, 10000, "SquaredExponential", 1e-6, theta, 3)
run("+ one time_shard (profile_schedule toggled)")
for it in range(25):
    mo.eval(np.tile(theta, (64, 1)) + 1e-3 * it, grad=True)
    mo.predict_variance_batch_dev(d_Xs.data_ptr(), 10000, d_mean.data_ptr(), d_var.data_ptr())
run("+ 25 bench steps (1.4 s of sustained load)")
run("... 32 emulators", 32)
run("... 16 emulators", 16)
s = bench.time_shard(M, GPPriors, 2, 2000, 10, 32, 10000, "SquaredExponential", 1e-6, theta, 7, map_starts=15)
print("time_shard(32) with its fit_GP_MAP: %.3f s (first %.3f) %.1f TF" % (s["fit_GP_MAP_s"], s["fit_GP_MAP_first_call_s"], s["fit_GP_MAP_TFLOPs"]), flush=True)
run("after time_shard(32)")
