// Device helpers shared by the covariance kernels (kernels_cov.hip) and the Cholesky update kernel that generates the
// covariance tile it touches first (kernels_gemm.hip): kernel functions, LDS staging of input rows, and the rule for
// the entries of the augmented matrix (targets / design rows / padding).
#pragma once
#include "launch.h"

namespace mogp {

template <int KT>
__device__ __forceinline__ double kern_val(double r2) {
  if (KT == 0) return exp(-0.5 * r2);
  const double s = sqrt(5.0 * r2);
  return (1.0 + s + (5.0 / 3.0) * r2) * exp(-s);
}
// dk/d(r2)
template <int KT>
__device__ __forceinline__ double kern_dr2(double r2) {
  if (KT == 0) return -0.5 * exp(-0.5 * r2);
  const double s = sqrt(5.0 * r2);
  return -(5.0 / 6.0) * (1.0 + s) * exp(-s);
}

// stage rows [r0, r0+64) of Xg (nrows, D) into sx[d*64 + r]; rows >= nrows are zero filled
__device__ __forceinline__ void stage_rows(const double* __restrict__ Xg, int nrows, int D, int r0, double* sx) {
  const int cnt = 64 * D;
  const int avail = max(0, min(64, nrows - r0)) * D;
  const double* src = Xg + (size_t)r0 * D;
  for (int e = threadIdx.x; e < cnt; e += 256) {
    const int r = e / D, d = e - r * D;
    sx[d * 64 + r] = (e < avail) ? src[e] : 0.0;
  }
}

// Entry (i, j) of the augmented matrix A: K + nugget I for i, j < n; rows n .. n+R-1 carry the right-hand sides (row n =
// targets, rows n+1.. = design-matrix columns of the analytic mean) with PAD_BIG on their diagonal; identity beyond.
//   sk = sigma^2 k(x_i, x_j) (only used for i, j < n)
__device__ __forceinline__ double cov_entry(const BatchView& v, const double* __restrict__ T, int i, int j, double sk, double nug) {
  const int n = v.n;
  const int hi = max(i, j), lo = min(i, j);
  if (hi < n) return (i == j) ? sk + nug : sk;
  if (hi < n + v.R) {
    if (lo < n) return (hi == n) ? T[lo] : v.H[(size_t)(hi - n - 1) * n + lo];
    return (lo == hi) ? PAD_BIG : 0.0;
  }
  return (i == j) ? 1.0 : 0.0;
}

// sigma^2-free kernel value of ONE pair from LDS-staged coordinates (si / sj as written by stage_rows), with exactly the
// operation order of micro_r2 / micro_k so that every entry is bit-identical wherever it is generated.
//   KT: 0 squared exponential, 1 Matern-5/2, 2 product of one-dimensional Matern-5/2
template <int KT>
__device__ __forceinline__ double pair_kval(const double* si, const double* sj, const double* __restrict__ P, int D, int row, int col) {
  if (KT < 2) {
    double r2 = 0.0;
    for (int d = 0; d < D; ++d) {
      const double df = si[d * 64 + row] - sj[d * 64 + col];
      r2 = __builtin_fma(P[d] * df, df, r2);
    }
    return kern_val<KT>(r2);
  }
  double k = 1.0, ssum = 0.0;
  for (int d = 0; d < D; ++d) {
    const double df = si[d * 64 + row] - sj[d * 64 + col];
    const double r2 = P[d] * df * df;
    const double sd = sqrt(5.0 * r2);
    k *= 1.0 + sd + (5.0 / 3.0) * r2;
    ssum += sd;
  }
  return k * exp(-ssum);
}

}  // namespace mogp
