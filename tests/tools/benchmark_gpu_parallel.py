"""The workload of the reference's own GPU timing harness (mogp_emulator/benchmarks/benchmark_gpu_parallel.py:1-150): 30
six-dimensional input points, 1 ... 32 emulators (targets of the reference's data file, kept as the fixture
tests/golden/timingtestdata.npz), `fit_GP_MAP` with its default 15 starts and a prediction at 30 points, wall-clock per
call.  --run_cpu adds the oracle's scipy L-BFGS-B multi-start fit on the host (one emulator after the other).
Prints a table (and writes --output_csv_filename)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M


def load_data(n_emulators):
    f = np.load(os.path.join(ROOT, "tests", "golden", "timingtestdata.npz"))
    assert f["inputs"].shape[0] == f["targets"].shape[1] and f["predict_points"].shape[1] == f["inputs"].shape[1]
    return f["inputs"], f["targets"][:n_emulators], f["predict_points"]


def run_single_test(n_emulators, use_gpu, inputs, targets, x_predict):
    if use_gpu:
        mgp = M.MultiOutputGP_GPU(inputs, targets)
        t0 = time.perf_counter()
        mgp = M.fit_GP_MAP(mgp)
        t1 = time.perf_counter()
        mgp.predict(x_predict)
        t2 = time.perf_counter()
        assert mgp.get_indices_not_fit() == []
    else:
        from oracle import cpu_ref as R
        from mogp_emulator_amd.Priors import GPPriors, InvGammaPrior
        dp = GPPriors.default_priors(inputs, inputs.shape[1], "adaptive")       # the priors both classes use by default
        corr = [R.Prior("invgamma", p.shape, p.scale) if isinstance(p, InvGammaPrior) else R.Prior() for p in dp.corr]
        gps = [R.GPRef(inputs, t, nugget="adaptive", priors=R.GPPriorsRef(inputs.shape[1], "adaptive", corr=corr)) for t in targets]
        t0 = time.perf_counter()
        gps = [R.fit_GP_MAP_ref(g, n_tries=15) for g in gps]
        t1 = time.perf_counter()
        for g in gps:
            if g.theta is not None:
                g.predict(x_predict, deriv=True)
        t2 = time.perf_counter()
    return t1 - t0, t2 - t1


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Run the reference's GPU timing workload")
    ap.add_argument("--num_reps", type=int, default=3)
    ap.add_argument("--max_num_emulators", type=int, choices=[2, 4, 8, 16, 32], default=32)
    ap.add_argument("--output_csv_filename")
    ap.add_argument("--run_cpu", action="store_true")
    args = ap.parse_args()
    rows = []
    for n_em in [2 ** k for k in range(int(np.log2(args.max_num_emulators)) + 1)]:
        inputs, targets, x_predict = load_data(n_em)
        for gpu in ([True] + ([False] if args.run_cpu else [])):
            for _ in range(args.num_reps if gpu else 1):
                ft, pt = run_single_test(n_em, gpu, inputs, targets, x_predict)
                rows.append((gpu, n_em, ft, pt))
    print("%5s %12s %12s %14s" % ("GPU", "n_emulators", "fit_time_s", "predict_time_s"))
    for gpu in (True, False):
        for n_em in sorted(set(r[1] for r in rows)):
            sel = [r for r in rows if r[0] == gpu and r[1] == n_em]
            if sel:
                print("%5s %12d %12.4f %14.5f" % (gpu, n_em, np.median([r[2] for r in sel]), np.median([r[3] for r in sel])))
    if args.output_csv_filename:
        with open(args.output_csv_filename, "w") as fh:
            fh.write("GPU,n_emulators,fit_time,predict_time\n")
            for r in rows:
                fh.write("%s,%d,%.6f,%.6f\n" % r)
