"""Context for the numbers in DESIGN.md / HISTORY.md: the same linear algebra through PyTorch-ROCm's vendor path (rocSOLVER / rocBLAS /
hipSOLVER via torch.linalg) on the benchmark shape -- 64 matrices of n=2000, fp64.  Not part of the product or the tests."""
import sys, time
import numpy as np
import torch

B, n, m = 64, 2000, 10000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
X = torch.rand(n, 10, dtype=torch.float64, device=dev)
d2 = torch.cdist(X, X) ** 2
K = torch.exp(-0.5 * d2 / (0.3 ** 2 * 10)) + 1e-6 * torch.eye(n, dtype=torch.float64, device=dev)
Kb = K.unsqueeze(0).repeat(B, 1, 1).contiguous()
t = torch.rand(B, n, 1, dtype=torch.float64, device=dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


try:
    dt, L = timed(lambda: torch.linalg.cholesky(Kb))
except Exception as e:
    print("torch.linalg.cholesky failed:", str(e).splitlines()[0]); sys.exit(0)
print("torch.linalg.cholesky (64 x 2000 x 2000, f64): %.1f ms  (%.1f TFLOP/s)" % (dt * 1e3, B * n ** 3 / 3 / dt / 1e12))
print("max |L L^T - K| = %.2e" % float((L[0] @ L[0].T - K).abs().max()), flush=True)
dt, Ki = timed(lambda: torch.cholesky_inverse(L))
print("torch.cholesky_inverse (K^-1 from L): %.1f ms  (%.1f TFLOP/s of 2n^3/3)" % (dt * 1e3, B * 2 * n ** 3 / 3 / dt / 1e12))
Ks = torch.rand(B, n, m // 2, dtype=torch.float64, device=dev)
dt, V = timed(lambda: torch.linalg.solve_triangular(L, Ks, upper=False), reps=2)
print("torch.linalg.solve_triangular L^-1 K*^T, %d columns: %.1f ms  (%.1f TFLOP/s of m n^2)" % (m // 2, dt * 1e3, B * (m // 2) * n * n / dt / 1e12))
Linv = torch.linalg.inv(L) if False else None
dt, G = timed(lambda: torch.bmm(L, Ks), reps=2)
print("torch.bmm (n x n) x (n x %d), full square: %.1f ms  (%.1f TFLOP/s of 2 m n^2)" % (m // 2, dt * 1e3, B * 2 * (m // 2) * n * n / dt / 1e12))
