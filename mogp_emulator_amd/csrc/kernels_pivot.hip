// Pivoted Cholesky for nugget="pivot" (linalg/cholesky.py:284-327, which calls LAPACK dpstrf): diagonal pivoting,
//
//   per column j:   w_i   = a_ii - sum_{k<j} l_ik^2 for the remaining rows, p = first arg max_i w_i
//                   stop when w_p <= n * eps * max_i a_ii  (rank = j)
//                   interchange rows / columns j <-> p, l_jj = sqrt(w_p)
//                   l_ij  = (a_ij - sum_{k<j} l_ik l_jk) / l_jj      for every row below (including the right-hand-side rows)
//
// organised like LAPACK's blocked dpstrf: block columns of 64.  Inside a block the pivot of step j depends on column
// j-1, so the steps are sequential and each is a matrix-vector product over the block's columns only (BLAS-2, one
// workgroup of sixteen waves per emulator: rows are contiguous, each wave owns sixteen rows at a time and its 64 lanes
// take one column each, so every load instruction moves one full 512-byte line and a wave has sixteen of them in flight);
// after the block, the rank-64 update of everything to its right is the MFMA update of the blocked Cholesky
// (launch_update_narrow / launch_update_trailing), which also keeps the diagonal that the next block's pivot search
// reads up to date.  The host drops an emulator from the batch when its factorisation stops early.
//
// Rank-deficient case, with the semantics of LAPACK's blocked dpstrf at its 64-column block size (= this blocking): the rows
// that were never chosen hold, below the diagonal, what the factorisation left there -- the interchanged input entries
// minus the rank-64 updates of the COMPLETED block columns (for n <= 64: the input entries, as the unblocked dpstf2) -- and the
// diagonal l_{r-1,r-1} / ((r+1)(r+2)...(i+1)); the forward substitution of the right-hand-side rows is continued through
// that block so that rows n.. of A hold L^-1 [t, H] for the complete factor (pstrf_tail_kernel, HISTORY.md section 3d).
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include "launch.h"
#include "cov_dev.h"

namespace mogp {

constexpr int PSTRF_THREADS = 1024;
constexpr int PSTRF_ROWS = 16;        // (rows16_sum is written for exactly sixteen)
//       // rows per wave in flight in the matrix-vector product (rows are 16 KB apart: latency bound)

// Sum over the 64 lanes, returned in every lane.  DPP moves inside the rows of 16 lanes (full-rate VALU, no LDS
// crossbar: the __shfl_xor butterfly cost ~1800 cycles per matrix row and bounded the whole panel kernel), then the four
// row totals through v_readlane.
template <int CTRL>
__device__ __forceinline__ double dpp_add(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
  return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double x) {
  x = dpp_add<0xB1>(x);     // quad_perm:[1,0,3,2]
  x = dpp_add<0x4E>(x);     // quad_perm:[2,3,0,1]
  x = dpp_add<0x141>(x);    // row_half_mirror
  x = dpp_add<0x140>(x);    // row_mirror
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (r0 + r1) + (r2 + r3);
}

// Sixteen per-row partial sums per lane -> the sixteen row totals, total of row (lane >> 2) in every lane: a
// reduce-scatter (each exchange halves the number of values a lane carries) instead of sixteen butterflies.  The two
// cross-row exchanges are gfx950's v_permlane32_swap / v_permlane16_swap, the rest DPP moves inside the rows of 16 lanes.
// 63 instead of ~400 VALU instructions per sixteen rows -- the butterflies were what bounded the panel kernel.
__device__ __forceinline__ double comb32(double a, double b) {      // lanes 0-31: a.lo + a.hi halves, lanes 32-63: b's
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double comb16(double a, double b) {      // rows (of 16 lanes) 0, 2: a's row pairs, rows 1, 3: b's
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int CTRL>
__device__ __forceinline__ double comb_dpp(double a, double b, bool upper) {   // lower partner keeps a, upper partner keeps b
  const double keep = upper ? b : a, send = upper ? a : b;
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(send), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(send), CTRL, 0xF, 0xF, true);
  return keep + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rows16_sum(const double (&s)[16], int lane) {
  double t[8], w[4];
#pragma unroll
  for (int u = 0; u < 8; ++u) t[u] = comb32(s[u], s[u + 8]);           // lane bit 5 <-> row bit 3
#pragma unroll
  for (int u = 0; u < 4; ++u) w[u] = comb16(t[u], t[u + 4]);           // lane bit 4 <-> row bit 2
  const bool b3 = lane & 8, b2 = lane & 4;
  const double z0 = comb_dpp<0x128>(w[0], w[2], b3), z1 = comb_dpp<0x128>(w[1], w[3], b3);   // row_ror:8, lane bit 3 <-> row bit 1
  double y = comb_dpp<0x141>(z0, z1, b2);                              // row_half_mirror, lane bit 2 <-> row bit 0
  y = dpp_add<0xB1>(y);
  return dpp_add<0x4E>(y);
}

// block-wide "first arg max" of (val, idx) over PSTRF_THREADS threads; NaN never wins; idx = none when nothing does
struct ArgMax {
  double val;
  int idx;
};
__device__ __forceinline__ ArgMax block_arg_max(double bv, int bi, double* s_val, int* s_idx) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = PSTRF_THREADS / 64;
#pragma unroll
  for (int off = 32; off; off >>= 1) {
    const double ov = __shfl_down(bv, off);
    const int oi = __shfl_down(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  // (the callers have a barrier between the reads below and the next call's writes)
  if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
  __syncthreads();
  bv = s_val[0];
  bi = s_idx[0];
  for (int w = 1; w < nw; ++w) {
    const double ov = s_val[w];
    const int oi = s_idx[w];
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  return ArgMax{bv, bi};
}

__device__ __forceinline__ double* pstrf_work(double* work, const BatchView& v, int emu) { return work + (size_t)emu * pstrf_work_doubles(v.NP); }

// rank[emu] = -1 (running), perm = identity, stopping threshold from the largest diagonal entry
__global__ __launch_bounds__(PSTRF_THREADS) void pstrf_begin_kernel(BatchView v, int* __restrict__ perm, int* __restrict__ rank,
                                                                     int* __restrict__ info, double* __restrict__ work) {
  __shared__ double s_val[PSTRF_THREADS / 64 + 1];
  __shared__ int s_idx[PSTRF_THREADS / 64 + 1];
  const int emu = v.idx ? v.idx[blockIdx.x] : blockIdx.x;
  const int n = v.n, ld = v.LD;
  const double* A = v.A + (size_t)emu * v.MS;
  int* P = perm + (size_t)emu * n;
  double bv = -INFINITY;
  int bi = n;
  for (int i = threadIdx.x; i < n; i += PSTRF_THREADS) {
    const double d = A[(size_t)i * ld + i];
    P[i] = i;
    if (d > bv) { bv = d; bi = i; }
  }
  const ArgMax m = block_arg_max(bv, bi, s_val, s_idx);
  if (threadIdx.x == 0) {
    const bool bad = m.idx >= n || !(m.val > 0.0);      // dpstf2: largest diagonal entry <= 0 or NaN -> rank 0, info 1
    rank[emu] = bad ? 0 : -1;
    info[emu] = bad ? 1 : 0;
    pstrf_work(work, v, emu)[2 * v.NP] = n * (0.5 * DBL_EPSILON) * m.val;     // tol < 0: N * DLAMCH('Epsilon') * max diagonal
  }
}

// columns [k0, k0 + jb), jb <= 64, of every running emulator in the launch
__global__ __launch_bounds__(PSTRF_THREADS) void pstrf_panel_kernel(BatchView v, int k0, int jb, int* __restrict__ perm,
                                                                     int* __restrict__ rank, double* __restrict__ work) {
  __shared__ double s_val[PSTRF_THREADS / 64 + 1];
  __shared__ int s_idx[PSTRF_THREADS / 64 + 1];
  const int emu = v.idx ? v.idx[blockIdx.x] : blockIdx.x;
  if (rank[emu] >= 0) return;
  const int n = v.n, ld = v.LD, nr = v.n + v.R;
  double* A = v.A + (size_t)emu * v.MS;
  double* dots = pstrf_work(work, v, emu);          // sum over the block's columns of l_ik^2
  double* diag = dots + v.NP;                       // diagonal as the earlier blocks' updates left it
  const double dstop = dots[2 * v.NP];
  int* P = perm + (size_t)emu * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = PSTRF_THREADS / 64;
  for (int i = k0 + tid; i < n; i += PSTRF_THREADS) {
    dots[i] = 0.0;
    diag[i] = A[(size_t)i * ld + i];
  }
  __syncthreads();
  int r = -1;
  // pivot candidate of the first column; for the later columns it comes out of the previous column's epilogue
  ArgMax cur;
  {
    double bv = -INFINITY;
    int bi = n;
    for (int i = k0 + tid; i < n; i += PSTRF_THREADS) {
      const double w = diag[i] - dots[i];
      if (w > bv) { bv = w; bi = i; }
    }
    cur = block_arg_max(bv, bi, s_val, s_idx);
  }
  for (int j = k0; j < k0 + jb; ++j) {
    const double piv = cur.val;
    const int p = cur.idx;
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
        rp[p] = diag[j];                   // the diagonal entries change places as well (a_jj itself becomes l_jj below)
        diag[p] = diag[j];
        const int q = P[j]; P[j] = P[p]; P[p] = q;
      }
    }
    const double ajj = sqrt(piv);
    if (tid == 0) rj[j] = ajj;
    __syncthreads();
    const double inv = 1.0 / ajj;            // dpstf2 scales the column by the reciprocal
    double bv = -INFINITY;                   // this thread's candidate for the next pivot (rows in increasing order)
    int bi = n;
    for (int i0 = j + 1 + wave * PSTRF_ROWS; i0 < nr; i0 += nw * PSTRF_ROWS) {
      const double* row[PSTRF_ROWS];
      double s[PSTRF_ROWS];
#pragma unroll
      for (int u = 0; u < PSTRF_ROWS; ++u) {
        row[u] = A + (size_t)min(i0 + u, nr - 1) * ld;
        s[u] = 0.0;
      }
      const int i = i0 + (lane >> 2);
      const bool owner = (lane & 3) == 0 && i < nr;          // this lane finishes row i
      double* c = A + (size_t)min(i, nr - 1) * ld + j;
      double cold = 0.0, dold = 0.0, gold = 0.0;
      if (owner) {
        cold = *c;
        if (i < n) { dold = dots[i]; gold = diag[i]; }
      }
      {
        // at most 63 earlier columns in this launch: one predicated load per row, all of them in flight together
        const int ka = k0 + lane;
        const bool pa = ka < j;
        const double xa = pa ? rj[ka] : 0.0;
        double va[PSTRF_ROWS];
#pragma unroll
        for (int u = 0; u < PSTRF_ROWS; ++u) va[u] = pa ? row[u][ka] : 0.0;
#pragma unroll
        for (int u = 0; u < PSTRF_ROWS; ++u) s[u] = va[u] * xa;
      }
      const double mine = rows16_sum(s, lane);          // total of row i0 + (lane >> 2)
      if (owner) {
        const double val = (cold - mine) * inv;
        *c = val;
        if (i < n) {
          const double dn = dold + val * val;
          dots[i] = dn;
          const double w = gold - dn;
          if (w > bv) { bv = w; bi = i; }
        }
      }
    }
    cur = block_arg_max(bv, bi, s_val, s_idx);          // its barrier also closes the column
  }
  if (tid == 0) {
    if (r >= 0) rank[emu] = r;
    else if (k0 + jb >= n) rank[emu] = n;
  }
}

// sigma^2 k(x_a, x_b) with the operation order of the covariance build (kernels_cov.hip micro_k)
__device__ __forceinline__ double cov_pair_global(const BatchView& v, const double* __restrict__ X, const double* __restrict__ P, int a, int b) {
  const int D = v.D;
  const double* xa = X + (size_t)a * D;
  const double* xb = X + (size_t)b * D;
  if (v.kernel_type < 2) {
    double r2 = 0.0;
    for (int d = 0; d < D; ++d) {
      const double sc = sqrt(P[d]);          // (the K build stages its inputs multiplied by sqrt(e_d): stage_rows<true>, micro_r2<.., true>)
      const double df = xa[d] * sc - xb[d] * sc;
      r2 = __builtin_fma(df, df, r2);
    }
    return P[D] * (v.kernel_type == 0 ? kern_val<0>(r2, EXP_TAB_G) : kern_val<1>(r2, EXP_TAB_G));
  }
  double k = 1.0, ssum = 0.0;
  for (int d = 0; d < D; ++d) {
    const double df = xa[d] - xb[d];
    const double r2 = P[d] * df * df;
    const double sd = sqrt(5.0 * r2);
    k *= 1.0 + sd + (5.0 / 3.0) * r2;
    ssum += sd;
  }
  return P[D] * (k * lean_exp_neg<false>(ssum, EXP_TAB_G));
}

// Emulators whose factorisation stopped at rank r < n: set the replacement diagonal of the block that was skipped and take the
// right-hand-side rows through it (regen: first put the INPUT entries back into that block -- A0 != null: from the input matrix
// (n x n, row-major) instead of the kernel function).
__global__ __launch_bounds__(PSTRF_THREADS) void pstrf_tail_kernel(BatchView v, const int* __restrict__ perm, const int* __restrict__ rank,
                                                                    const double* __restrict__ X0, const double* __restrict__ A0, int regen) {
  const int emu = v.idx ? v.idx[blockIdx.x] : blockIdx.x;
  const int n = v.n, ld = v.LD, r = rank[emu];
  if (r <= 0 || r >= n) return;
  double* A = v.A + (size_t)emu * v.MS;
  const int* P = perm + (size_t)emu * n;
  const double* prm = v.P ? v.P + (size_t)emu * v.PS : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = n - r;
  // The block that was skipped keeps what the factorisation left in it, as with LAPACK's blocked dpstrf (64-column blocks, as here):
  // the interchanged input entries MINUS the rank-64 updates of the completed block columns -- for n <= 64 the input entries
  // themselves (the golden vectors at n = 43), beyond that the Schur complement with respect to the completed blocks, which for
  // rows skipped as repeats of earlier pivots is rounding residue.  (Rounds 1-3 put the input entries back for every n, the
  // semantics of the unblocked dpstf2: with two skipped rows the forward substitution then multiplies the first one's amplified
  // rounding residue by an O(1) entry and divides by a replacement diagonal ~1e-8 -- log-posteriors off by up to 0.11 relative
  // against scipy's dpstrf in the randomised test, n = 400, two repeated points.)  MOGP_PIVOT_TAIL=input restores that.
  if (regen) {
    for (long e = tid; e < (long)m * m; e += PSTRF_THREADS) {
      const int i = r + (int)(e / m), j = r + (int)(e % m);
      if (j >= i) continue;
      A[(size_t)i * ld + j] = A0 ? A0[(size_t)P[i] * n + P[j]] : cov_pair_global(v, X0, prm, P[i], P[j]);
    }
  }
  for (int e = tid; e < v.R * m; e += PSTRF_THREADS) {
    const int c = e / m, j = r + e % m;
    A[(size_t)(n + c) * ld + j] = (c == 0) ? v.T[(size_t)emu * n + P[j]] : v.H[(size_t)(c - 1) * n + P[j]];
  }
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
  // one wave per right-hand-side row, sequential in j (column j needs the row's entries of all earlier columns)
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
__global__ void pstrf_end_kernel(BatchView v) {
  const int emu = v.idx ? v.idx[blockIdx.x] : blockIdx.x;
  double* A = v.A + (size_t)emu * v.MS;
  const int R = v.R, n = v.n;
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) {
    const int a = e / R, b = e % R;
    if (b <= a) A[(size_t)(n + a) * v.LD + n + b] = (a == b) ? sqrt(PAD_BIG) : 0.0;
  }
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

void launch_pstrf_begin(const BatchView& v, int* perm, int* rank, int* info, double* work, hipStream_t s) {
  hipLaunchKernelGGL(pstrf_begin_kernel, dim3(v.nb), dim3(PSTRF_THREADS), 0, s, v, perm, rank, info, work);
}

void launch_pstrf_panel(const BatchView& v, int k0, int jb, int* perm, int* rank, double* work, hipStream_t s) {
  hipLaunchKernelGGL(pstrf_panel_kernel, dim3(v.nb), dim3(PSTRF_THREADS), 0, s, v, k0, jb, perm, rank, work);
}

void launch_pstrf_tail(const BatchView& v, const int* perm, const int* rank, const double* X0, const double* A0, hipStream_t s) {
  static const int regen = [] { const char* e = getenv("MOGP_PIVOT_TAIL"); return (e && e[0] == 'i') ? 1 : 0; }();
  hipLaunchKernelGGL(pstrf_tail_kernel, dim3(v.nb), dim3(PSTRF_THREADS), 0, s, v, perm, rank, X0, A0, regen);
}

void launch_pstrf_end(const BatchView& v, hipStream_t s) {
  if (v.R > 0) hipLaunchKernelGGL(pstrf_end_kernel, dim3(v.nb), dim3(64), 0, s, v);
}

void launch_permute_rows(const BatchView& v, const double* X, const int* perm, double* Xp, hipStream_t s) {
  hipLaunchKernelGGL(permute_rows_kernel, dim3((v.n * v.D + 255) / 256, v.nb), dim3(256), 0, s, v, X, perm, Xp);
}

}  // namespace mogp
