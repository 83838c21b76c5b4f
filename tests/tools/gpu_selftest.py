"""Stage-by-stage check of the HIP path against the oracle (development aid; prints max errors)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mogp_emulator_amd import _capi as C
from oracle import cpu_ref as R

lib = C.load()
print("device ok:", lib.mogp_have_compatible_device(), lib.mogp_version())


def relerr(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def run_case(name, X, t, theta, kern, nugget, Xs):
    n, D = X.shape
    X = np.ascontiguousarray(X, dtype=np.float64); t = np.ascontiguousarray(t, dtype=np.float64)
    kt = 0 if kern == R.SQEXP else 1
    if isinstance(nugget, str):
        nt, ns = (0, 0.) if nugget == "adaptive" else (1, 0.)
    else:
        nt, ns = 2, float(nugget)
    h = lib.mogp_densegp_create(C.dptr(X), n, D, C.dptr(t), 100000, None, kt, nt, ns)
    assert h, C.last_error()
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    t0 = time.time()
    C.check(lib.mogp_densegp_fit(h, C.dptr(theta), len(theta)))
    lp = np.zeros(1); C.check(lib.mogp_densegp_get_logpost(h, C.dptr(theta), len(theta), C.dptr(lp)))
    gp = R.GPRef(X, t, kernel=kern, nugget=nugget)
    ref_lp = gp.fit(theta)
    K = np.zeros((n, n)); C.check(lib.mogp_densegp_get_K(h, C.dptr(K)))
    L = np.zeros((n, n)); C.check(lib.mogp_densegp_get_cholesky_lower(h, C.dptr(L))); L = np.tril(L.T)
    a = np.zeros(n); C.check(lib.mogp_densegp_get_invQt(h, C.dptr(a)))
    g = np.zeros(len(theta)); C.check(lib.mogp_densegp_logpost_deriv(h, C.dptr(g), len(theta)))
    Q = np.zeros((n, n)); C.check(lib.mogp_densegp_get_invQ(h, C.dptr(Q)))
    m = Xs.shape[0]
    Xs = np.ascontiguousarray(Xs)
    mu = np.zeros(m); var = np.zeros(m); dv = np.zeros((m, D))
    C.check(lib.mogp_densegp_predict_variance_batch(h, C.dptr(Xs), m, D, C.dptr(mu), C.dptr(var), m))
    C.check(lib.mogp_densegp_predict_deriv(h, C.dptr(Xs), m, D, C.dptr(dv), m, D))
    rmu, rvar, rd = gp.predict(Xs, include_nugget=False, deriv=True)
    Kref = gp.get_K_matrix()
    Qref = np.linalg.inv(Kref + gp.nugget * np.eye(n)) if n <= 600 else None
    print("[%s] n=%d D=%d kern=%s nugget=%s  (%.2fs)" % (name, n, D, kern, nugget, time.time() - t0))
    print("   nugget gpu %.6e ref %.6e" % (lib.mogp_densegp_get_nugget_size(h), gp.nugget))
    print("   logpost gpu %.12e ref %.12e rel %.2e" % (lp[0], ref_lp, abs(lp[0] - ref_lp) / abs(ref_lp)))
    print("   K    rel %.2e" % relerr(K, Kref))
    print("   L    rel %.2e" % relerr(L, gp.L))
    print("   alpha rel %.2e" % relerr(a, gp.Kinv_t))
    print("   grad rel %.2e  gpu %s ref %s" % (relerr(g, gp.logpost_deriv(theta)), g[:3], gp.logpost_deriv(theta)[:3]))
    if Qref is not None:
        print("   invQ rel %.2e" % relerr(Q, Qref))
    print("   mean rel %.2e  var abs %.2e (sig2=%.2e) deriv rel %.2e" % (relerr(mu, rmu), np.max(np.abs(var - rvar)), np.exp(theta[D]), relerr(dv, rd)))
    lib.mogp_densegp_destroy(h)


G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "fixture_2x3.npz"))
run_case("2x3", g["X"], g["t"], np.ones(4), R.SQEXP, 0., g["Xs"])
run_case("2x3m", g["X"], g["t"], np.ones(4), R.MAT52, 0., g["Xs"])
g = np.load(os.path.join(G, "grid11.npz"))
for kern in (R.SQEXP, R.MAT52):
    run_case("grid11", g["X"], g["t"], [-1., -1., -2.], kern, 1e-6, g["Xs"])
    run_case("grid11", g["X"], g["t"], [-1., -1., -2., np.log(1e-6)], kern, "fit", g["Xs"])
    run_case("grid11", g["X"], g["t"], [-1., -1., -2.], kern, "adaptive", g["Xs"])
g = np.load(os.path.join(G, "c1_n200_d4.npz"))
run_case("c1", g["X"], g["T"][0], g["SquaredExponential_fixed_theta"], R.SQEXP, 1e-6, g["Xs"])
run_case("c1", g["X"], g["T"][0], g["Matern52_fit_theta"], R.MAT52, "fit", g["Xs"])
g = np.load(os.path.join(G, "n500_d10.npz"))
run_case("n500", g["X"], g["T"][0], g["SquaredExponential_fixed_theta"], R.SQEXP, 1e-6, g["Xs"])
run_case("n500", g["X"], g["T"][0], g["Matern52_fit_theta"], R.MAT52, "fit", g["Xs"])
# a bigger one (several outer blocks, ragged)
rng = np.random.default_rng(1)
n, D = 1500, 10
X = rng.uniform(0, 1, (n, D)); w = rng.normal(size=D)
t = np.sin(2 * np.pi * X @ w / np.sqrt(D)) + 0.01 * rng.normal(size=n)
th = np.array([-2 * np.log(0.3 * np.sqrt(D))] * D + [0.])
run_case("n1500", X, t, th, R.SQEXP, 1e-6, rng.uniform(0, 1, (300, D)))
