// Pivoted Cholesky for nugget="pivot" (linalg/cholesky.py:284-327, which calls LAPACK dpstrf; with the block size
// reference LAPACK selects for it that is the unblocked dpstf2): a left-looking column algorithm with diagonal pivoting.
//
//   per column j:   w_i   = a_ii - sum_{k<j} l_ik^2 for the remaining rows, p = first arg max_i w_i
//                   stop when w_p <= n * eps * max_i a_ii  (rank = j)
//                   interchange rows / columns j <-> p, l_jj = sqrt(w_p)
//                   l_ij  = (a_ij - sum_{k<j} l_ik l_jk) / l_jj      for every row below (including the right-hand-side rows)
//
// The pivot of step j depends on column j-1, so the n steps are sequential and the work per step is a matrix-vector
// product: this is HBM/L2-bound BLAS-2 work (8 n^3 / 6 bytes per emulator), not MFMA work, and one workgroup of sixteen
// waves runs a whole emulator -- the batch over emulators is what fills the chip.  Rows are contiguous (row-major lower
// triangle), so each wave owns rows and its lanes stride along k: every load instruction moves full 512-byte lines.
//
// Rank-deficient case, as the reference has it: the rows that were never chosen keep the (interchanged) input entries
// below the diagonal and get the diagonal l_{r-1,r-1} / ((r+1)(r+2)...(i+1)); the forward substitution of the
// right-hand-side rows is continued through that block so that rows n.. of A hold L^-1 [t, H] for the complete factor.
#include <cfloat>
#include <cmath>
#include "launch.h"

namespace mogp {

constexpr int PSTRF_THREADS = 1024;
constexpr int PSTRF_ROWS = 4;        // rows per wave in flight in the matrix-vector product

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off; off >>= 1) x += __shfl_xor(x, off);
  return x;
}

__global__ __launch_bounds__(PSTRF_THREADS) void pstrf_kernel(BatchView v, int* __restrict__ perm, int* __restrict__ rank_out,
                                                               int* __restrict__ info, double* __restrict__ work) {
  __shared__ double s_val[PSTRF_THREADS / 64];
  __shared__ int s_idx[PSTRF_THREADS / 64];
  __shared__ double s_piv;
  __shared__ int s_p;
  const int emu = v.idx ? v.idx[blockIdx.x] : blockIdx.x;
  const int n = v.n, ld = v.LD, nr = v.n + v.R;
  double* A = v.A + (size_t)emu * v.MS;
  double* dots = work + (size_t)emu * 2 * v.NP;      // sum_k l_ik^2 so far
  double* diag = dots + v.NP;                         // diagonal of the (interchanged) input matrix
  int* P = perm + (size_t)emu * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = PSTRF_THREADS / 64;

  // block-wide "first arg max" of (val, idx); NaN never wins; idx = n when nothing does
  auto arg_max = [&](double bv, int bi) {
#pragma unroll
    for (int off = 32; off; off >>= 1) {
      const double ov = __shfl_down(bv, off);
      const int oi = __shfl_down(bi, off);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < nw; ++w)
        if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
      s_piv = bv;
      s_p = bi;
    }
    __syncthreads();
  };

  double bv = -INFINITY;
  int bi = n;
  for (int i = tid; i < n; i += PSTRF_THREADS) {
    const double d = A[(size_t)i * ld + i];
    dots[i] = 0.0;
    diag[i] = d;
    P[i] = i;
    if (d > bv) { bv = d; bi = i; }
  }
  arg_max(bv, bi);
  const double amax = s_piv;
  if (s_p >= n || !(amax > 0.0)) {          // dpstf2: largest diagonal entry <= 0 or NaN -> rank 0, info 1
    if (tid == 0) { rank_out[emu] = 0; info[emu] = 1; }
    return;
  }
  const double dstop = n * (0.5 * DBL_EPSILON) * amax;     // tol < 0: N * DLAMCH('Epsilon') * max diagonal

  int r = n;
  for (int j = 0; j < n; ++j) {
    bv = -INFINITY;
    bi = n;
    for (int i = j + tid; i < n; i += PSTRF_THREADS) {
      const double w = diag[i] - dots[i];
      if (w > bv) { bv = w; bi = i; }
    }
    arg_max(bv, bi);
    const double piv = s_piv;
    const int p = s_p;
    if (p >= n || !(piv > dstop)) { r = j; break; }
    double* rj = A + (size_t)j * ld;
    if (p != j) {
      double* rp = A + (size_t)p * ld;
      for (int k = tid; k < j; k += PSTRF_THREADS) { const double t = rj[k]; rj[k] = rp[k]; rp[k] = t; }
      for (int i = j + 1 + tid; i < p; i += PSTRF_THREADS) {
        double* c = A + (size_t)i * ld + j;
        const double t = *c; *c = rp[i]; rp[i] = t;
      }
      for (int i = p + 1 + tid; i < nr; i += PSTRF_THREADS) {
        double* row = A + (size_t)i * ld;
        const double t = row[j]; row[j] = row[p]; row[p] = t;
      }
      if (tid == 0) {
        double t = dots[j]; dots[j] = dots[p]; dots[p] = t;
        t = diag[j]; diag[j] = diag[p]; diag[p] = t;
        const int q = P[j]; P[j] = P[p]; P[p] = q;
      }
    }
    const double ajj = sqrt(piv);
    if (tid == 0) rj[j] = ajj;
    __syncthreads();
    const double inv = 1.0 / ajj;            // dpstf2 scales the column by the reciprocal
    for (int i0 = j + 1 + wave * PSTRF_ROWS; i0 < nr; i0 += nw * PSTRF_ROWS) {
      const double* row[PSTRF_ROWS];
      double s[PSTRF_ROWS];
#pragma unroll
      for (int u = 0; u < PSTRF_ROWS; ++u) {
        row[u] = A + (size_t)min(i0 + u, nr - 1) * ld;
        s[u] = 0.0;
      }
      for (int k = lane; k < j; k += 64) {
        const double x = rj[k];
#pragma unroll
        for (int u = 0; u < PSTRF_ROWS; ++u) s[u] = __builtin_fma(row[u][k], x, s[u]);
      }
      double mine = 0.0;
#pragma unroll
      for (int u = 0; u < PSTRF_ROWS; ++u) {
        const double t = wave_sum(s[u]);
        if (lane == u) mine = t;
      }
      const int i = i0 + lane;
      if (lane < PSTRF_ROWS && i < nr) {
        double* c = A + (size_t)i * ld + j;
        const double val = (*c - mine) * inv;
        *c = val;
        if (i < n) dots[i] += val * val;
      }
    }
    __syncthreads();
  }

  if (r < n) {
    __syncthreads();
    if (tid == 0) {
      // linalg/cholesky.py:321-325: L[i][i] = L[r-1][r-1] / cumprod(r+1 .. i+1)
      const double d = A[(size_t)(r - 1) * ld + (r - 1)];
      double div = 1.0;
      for (int i = r; i < n; ++i) {
        div *= (double)(i + 1);
        A[(size_t)i * ld + i] = d / div;
      }
    }
    __syncthreads();
    // forward substitution of the right-hand-side rows through the columns that were not factored: one wave per row,
    // sequential in j (column j needs the row's entries of all earlier columns)
    if (wave < v.R) {
      double* row = A + (size_t)(n + wave) * ld;
      for (int j = r; j < n; ++j) {
        const double* rj = A + (size_t)j * ld;
        double s = 0.0;
        for (int k = lane; k < j; k += 64) s = __builtin_fma(row[k], rj[k], s);
        s = wave_sum(s);
        if (lane == 0) row[j] = (row[j] - s) / rj[j];
        __threadfence_block();
      }
    }
  }
  // the right-hand-side rows close the augmented factor like the blocked path does: diagonal sqrt(PAD_BIG), zeros between
  if (tid < v.R) A[(size_t)(n + tid) * ld + (n + tid)] = sqrt(PAD_BIG);
  if (tid == 0) { rank_out[emu] = r; info[emu] = 0; }
}

__global__ __launch_bounds__(256) void permute_rows_kernel(BatchView v, const double* __restrict__ X, const int* __restrict__ perm,
                                                           double* __restrict__ Xp) {
  const int emu = v.idx ? v.idx[blockIdx.y] : blockIdx.y;
  const int n = v.n, D = v.D;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n * D) return;
  const int pos = e / D, d = e - pos * D;
  Xp[(size_t)emu * n * D + e] = X[(size_t)perm[(size_t)emu * n + pos] * D + d];
}

void launch_pstrf(const BatchView& v, int* perm, int* rank, int* info, double* work, hipStream_t s) {
  hipLaunchKernelGGL(pstrf_kernel, dim3(v.nb), dim3(PSTRF_THREADS), 0, s, v, perm, rank, info, work);
}

void launch_permute_rows(const BatchView& v, const double* X, const int* perm, double* Xp, hipStream_t s) {
  hipLaunchKernelGGL(permute_rows_kernel, dim3((v.n * v.D + 255) / 256, v.nb), dim3(256), 0, s, v, X, perm, Xp);
}

}  // namespace mogp
