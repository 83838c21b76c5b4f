"""Where does the time of a small fit go outside its kernels?  Wall time per evaluation against the sum of the HIP-event kernel times
(mogp_profile_*), for one tiny matrix (launch + synchronisation floor), C2 (one n = 2000) and the 8-emulator shard.
usage: python tools/host_gap.py"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M                        # noqa: E402
from mogp_emulator_amd import _capi                   # noqa: E402
from mogp_emulator_amd.Priors import GPPriors        # noqa: E402
from bench import synth                              # noqa: E402

lib = _capi.load()
TAGS = ("cov_build", "mchol", "chol_diag128", "chol_update", "chol_trsm128", "backsolve")
for B, n, d in ((1, 100, 4), (1, 2000, 10), (8, 2000, 10), (64, 2000, 10)):
    X, T, _ = synth(2, n, d, B, 8)
    gp = M.MultiOutputGP_GPU(X, T, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
    mo = gp._mogp_gpu
    th = np.tile(np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.]), (B, 1))
    for _ in range(5):
        mo.eval(th, grad=False)
    reps = 200 if n <= 2000 and B <= 8 else 40
    t0 = time.perf_counter()
    for it in range(reps):
        mo.eval(th, grad=False)
    wall = (time.perf_counter() - t0) / reps * 1e3
    lib.mogp_profile_reset(); lib.mogp_profile_enable(1)
    for it in range(20):
        mo.eval(th, grad=False)
    lib.mogp_profile_enable(0)
    tot, parts = 0.0, []
    ms, cnt, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
    for tag in TAGS:
        if lib.mogp_profile_get(tag.encode(), ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by)) == 0 and cnt.value:
            per = ms.value / 20.0
            tot += per
            parts.append("%s %.4f" % (tag, per))
    print("B=%d n=%d: wall %.4f ms per evaluation; tagged kernels %.4f ms (%s); outside them %.4f ms" % (B, n, wall, tot, ", ".join(parts), wall - tot), flush=True)
