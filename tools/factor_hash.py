"""sha256 of factors, alpha and log posterior of a few fits -- to check that two builds of libmogp_hip.so (MOGP_LIB_PATH) or two settings of a
switch give the same BITS.  usage: [MOGP_LIB_PATH=...] python tools/factor_hash.py"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mogp_emulator_amd as M                      # noqa: E402
from test_gpu_parity import synth, weak            # noqa: E402

CASES = [("SquaredExponential", 16, 2000, 10), ("Matern52", 16, 2000, 10), ("SquaredExponential", 64, 1000, 7), ("Matern52", 24, 1990, 3),
         ("SquaredExponential", 5, 4033, 40), ("SquaredExponential", 40, 1217, 1), ("SquaredExponential", 8, 2000, 10), ("SquaredExponential", 1, 3000, 4)]
for kern, B, n, d in CASES:
    X, T, _ = synth(4000 + n + B, n, d, B, 4)
    theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
    mo = M.MultiOutputGP_GPU(X, T, kernel=kern, nugget=1e-6, priors=weak(d, 1e-6))
    mo.fit(np.tile(theta, (B, 1)))
    h = hashlib.sha256()
    for k in sorted({0, B // 2, B - 1}):
        e = mo.emulators[k]
        h.update(np.ascontiguousarray(np.tril(e.L)).tobytes())
        h.update(np.ascontiguousarray(e.Kinv_t).tobytes())
        h.update(np.float64(e.current_logpost).tobytes())
    print("CASE", kern, B, n, d, h.hexdigest()[:32], repr(float(mo.emulators[0].current_logpost)))
