"""The reference's tsunami benchmark (mogp_emulator/benchmarks/benchmark_tsunami.py:1-97; its data file is the fixture
tests/golden/tsunamidata.npz: 210 simulations, 14 inputs, wave heights at up to 64 locations): `fit_GP_MAP` with the default
15 starts for 8 / 16 / 32 / 64 outputs, wall-clock and time per emulator.  The reference quotes "roughly 1 second per
emulator" for this on one core of a quad-core laptop.  --cpu N times the oracle's scipy multi-start fit for the first N outputs."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--cpu", type=int, default=0)
args = ap.parse_args()
f = np.load(os.path.join(ROOT, "tests", "golden", "tsunamidata.npz"))
inputs, targets = f["inputs"], f["targets"]
print("Num. Emulators    Execution Time (s)   Execution Time per Emulator (s)   all fit   sum logpost")
for n_em in (8, 16, 32, 64):
    best = None
    for rep in range(2):
        gp = M.MultiOutputGP_GPU(inputs, targets[:n_em])
        t0 = time.perf_counter()
        gp = M.fit_GP_MAP(gp)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    lp = sum(em.current_logpost for em in gp.emulators)
    print("%-18d%-21.4f%-34.5f%-10s%.4f" % (n_em, best, best / n_em, gp.get_indices_not_fit() == [], lp), flush=True)
if args.cpu:
    from oracle import cpu_ref as R
    from mogp_emulator_amd.Priors import GPPriors, InvGammaPrior
    dp = GPPriors.default_priors(inputs, inputs.shape[1], "adaptive")
    corr = [R.Prior("invgamma", p.shape, p.scale) if isinstance(p, InvGammaPrior) else R.Prior() for p in dp.corr]
    t0 = time.perf_counter()
    lps = []
    for k in range(args.cpu):
        g = R.fit_GP_MAP_ref(R.GPRef(inputs, targets[k], nugget="adaptive", priors=R.GPPriorsRef(inputs.shape[1], "adaptive", corr=corr)), n_tries=15)
        lps.append(g.current_logpost if g.theta is not None else float("nan"))
    dt = time.perf_counter() - t0
    print("host (oracle, scipy L-BFGS-B, %d emulators one after the other): %.2f s = %.2f s per emulator; logposts %s" % (
        args.cpu, dt, dt / args.cpu, np.round(lps, 4)))
    gp = M.fit_GP_MAP(M.MultiOutputGP_GPU(inputs, targets[:args.cpu]))
    print("device logposts of the same emulators: %s" % np.round([em.current_logpost for em in gp.emulators], 4))
