"""CPU baseline helper (test infrastructure, not product): the reference's parallel model for many outputs --
a process pool over emulators with ONE BLAS thread per worker (mogp_emulator/fitting.py:333-335,
MultiOutputGP.py:103-104) -- timed on the oracle.  Run as a script by bench.py's cpu_baseline leg:

    python oracle/pool_fit.py <config_id> <n> <d> <outputs> <workers> [<fits>]

<fits> (default <outputs>) objective evaluations are timed, cycling through the emulators -- with more workers than
outputs every worker still gets work.  Prints one JSON line {"wall_s", "fits", "workers", "mean_fit_s"}.  The synthetic data is regenerated here with
bench.synth (same seed), so nothing but five integers crosses the process boundary.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


_LIMIT = None


def _init():
    # bench.py also exports OMP/OPENBLAS/MKL_NUM_THREADS=1 before starting this script; this covers direct use
    global _LIMIT
    try:
        from threadpoolctl import threadpool_limits
        _LIMIT = threadpool_limits(1)
    except Exception:
        pass


def _fit(args):
    X, t, theta, nugget = args
    from oracle import cpu_ref as R
    t0 = time.perf_counter()
    R.GPRef(X, t, nugget=nugget).fit(theta)
    return time.perf_counter() - t0


def main():
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):     # inherited by the spawned workers
        os.environ[var] = "1"
    import multiprocessing as mp
    import numpy as np
    from bench import synth
    cid, n, d, B, workers = (int(a) for a in sys.argv[1:6])
    fits = int(sys.argv[6]) if len(sys.argv) > 6 else B
    X, T, _ = synth(cid, n, d, B, 8)
    theta = np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.])
    tasks = [(X, T[k % B], theta, 1e-6) for k in range(fits)]
    with mp.get_context("spawn").Pool(processes=workers, initializer=_init) as pool:
        pool.map(_fit, (tasks * workers)[:workers], chunksize=1)   # imports / first-touch outside the timed region
        t0 = time.perf_counter()
        times = pool.map(_fit, tasks, chunksize=1)
        wall = time.perf_counter() - t0
    print(json.dumps({"wall_s": wall, "fits": fits, "workers": workers, "mean_fit_s": float(np.mean(times))}))


if __name__ == "__main__":
    main()
