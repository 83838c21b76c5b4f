// Device helpers shared by the covariance kernels (kernels_cov.hip) and the pivoted Cholesky (kernels_pivot.hip): kernel
// functions with their exponential, LDS staging of input rows, and the rule for the entries of the augmented matrix
// (targets / design rows / padding).
#pragma once
#include "launch.h"
#include "exp_dev.h"

namespace mogp {

// 2^(j/256): the exponential's table.  The tiled kernels copy it into LDS (stage_exp_tab); one-thread-per-pair kernels read it here.
static __device__ const double EXP_TAB_G[256] = {MOGP_EXP_TAB_VALUES};

__device__ __forceinline__ void stage_exp_tab(double* tab) {     // 256-thread workgroups; the caller's next barrier publishes it
  if (threadIdx.x < 256) tab[threadIdx.x] = EXP_TAB_G[threadIdx.x];
}

template <int KT>
__device__ __forceinline__ double kern_val(double r2, const double* tab) {
  if (KT == 0) return lean_exp_neg<true>(r2, tab);
  const double s = sqrt(5.0 * r2);
  return (1.0 + s + (5.0 / 3.0) * r2) * lean_exp_neg<false>(s, tab);
}
// dk/d(r2)
template <int KT>
__device__ __forceinline__ double kern_dr2(double r2, const double* tab) {
  if (KT == 0) return -0.5 * lean_exp_neg<true>(r2, tab);
  const double s = sqrt(5.0 * r2);
  return -(5.0 / 6.0) * (1.0 + s) * lean_exp_neg<false>(s, tab);
}

// stage rows [r0, r0+64) of Xg (nrows, D) into sx[d*64 + r]; rows >= nrows are zero filled.  SC: multiplied by sqrt(P[d]) on the way
// (micro_r2<.., SC>; equal inputs stay equal, so r2 of repeated points is still exactly 0)
template <bool SC = false>
__device__ __forceinline__ void stage_rows(const double* __restrict__ Xg, int nrows, int D, int r0, double* sx, const double* __restrict__ P = nullptr) {
  const int cnt = 64 * D;
  const int avail = max(0, min(64, nrows - r0)) * D;
  const double* src = Xg + (size_t)r0 * D;
  for (int e = threadIdx.x; e < cnt; e += 256) {
    const int r = e / D, d = e - r * D;
    double x = (e < avail) ? src[e] : 0.0;
    if (SC) x *= sqrt(P[d]);
    sx[d * 64 + r] = x;
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

}  // namespace mogp
