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
def run(tag):
    b0, p0 = bench.counter("replica_engine_build_us"), bench.counter("replica_pool_us")
    r = bench.time_fit_map(M, X, T, "SquaredExponential", 1e-6, 15, 10)
    print("%-60s fit_GP_MAP %.3f s  %.1f TF  (replica engine built %.3f s, pool %.3f s, rest = final refit + freeing the replicas %.3f s)" % (
        tag, r["fit_GP_MAP_s"], r["fit_GP_MAP_TFLOPs"], (bench.counter("replica_engine_build_us") - b0) / 1e6, (bench.counter("replica_pool_us") - p0) / 1e6,
        r["fit_GP_MAP_s"] - (bench.counter("replica_engine_build_us") - b0) / 1e6 - (bench.counter("replica_pool_us") - p0) / 1e6), flush=True)
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
