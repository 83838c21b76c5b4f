// Panel kernels of the blocked right-looking Cholesky, the triangular solves and the
// triangular-inverse leaf.  These replace cusolverDnDpotrf / Dpotrs as used by the reference
// (densegp_gpu.hpp:451-474, 576-591) and the cub log-diagonal reduction (util.cu:38-49).
//
// Numerics: the factorisation itself uses exact substitution inside every 64-wide panel
// (no inverted diagonal blocks), i.e. the same backward-stable recurrence as LAPACK dpotrf /
// dtrsm.  A pivot that is not > 0 (or is NaN) marks the emulator as failed (info = 1-based
// column), which is what drives the adaptive-nugget ladder (linalg/cholesky.py:234-281).
#include "launch.h"

namespace mogp {

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int slot_emu(const int* idx, int z) { return idx ? idx[z] : z; }

// ---------------------------------------------------------------------------------------------
// potf2: unblocked Cholesky of the 64x64 diagonal block at (c0, c0), one workgroup per emulator.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void potf2_kernel(BatchView v, int c0, int* __restrict__ info) {
  __shared__ double S[64][65];
  __shared__ int fail;
  const int emu = slot_emu(v.idx, blockIdx.x);
  double* A = v.A + (size_t)emu * v.NP * v.NP + (size_t)c0 * v.NP + c0;
  const int ld = v.NP;
  const int t = threadIdx.x;
  for (int e = t; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    S[r][c] = A[(size_t)r * ld + c];
  }
  if (t == 0) fail = 0;
  __syncthreads();
  for (int j = 0; j < 64; ++j) {
    double d = S[j][j];
    if (!(d > 0.0)) {           // also catches NaN
      if (t == 0 && fail == 0) fail = j + 1;
      d = 1.0;
    }
    const double dj = sqrt(d);
    __syncthreads();            // everyone has read S[j][j]
    if (t > j && t < 64) S[t][j] = S[t][j] / dj;
    if (t == j) S[j][j] = dj;
    __syncthreads();
    // rank-1 update of the trailing lower triangle: (r, c) with r >= c > j
    const int m = 63 - j;       // trailing size
    for (int e = t; e < m * m; e += 256) {
      const int r = j + 1 + e / m, c = j + 1 + e % m;
      if (c <= r) S[r][c] -= S[r][j] * S[c][j];
    }
    __syncthreads();
  }
  for (int e = t; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (c <= r) A[(size_t)r * ld + c] = S[r][c];
  }
  if (t == 0 && fail != 0 && info[emu] == 0) info[emu] = c0 + fail;
}

// ---------------------------------------------------------------------------------------------
// Forward substitution with a 64x64 lower-triangular block held in LDS (row-major, ld 64):
//   solve  Lb * x = rhs  with x, rhs in registers (fully unrolled, broadcast LDS reads).
// Used per matrix row by the panel TRSM (x L^T = a  <=>  L x^T = a^T) and per unit vector by the
// triangular-inverse leaf.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void forward_subst_64(const double* __restrict__ Lb, double (&x)[64]) {
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    double s = x[c];
#pragma unroll
    for (int p = 0; p < c; ++p) s = __builtin_fma(-x[p], Lb[c * 64 + p], s);
    x[c] = s / Lb[c * 64 + c];
  }
}

// rows [r0, NP) of the column block [c0, c0+64):  X * L_kk^T = A_panel, one thread per row.
__global__ __launch_bounds__(256) void trsm_kernel(BatchView v, int c0, int r0) {
  __shared__ __attribute__((aligned(16))) double Lb[64 * 64];
  const int emu = slot_emu(v.idx, blockIdx.y);
  const int ld = v.NP;
  double* A = v.A + (size_t)emu * ld * ld;
  const int t = threadIdx.x;
  for (int e = t; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    Lb[e] = (c <= r) ? A[(size_t)(c0 + r) * ld + c0 + c] : 0.0;
  }
  __syncthreads();
  const int row = r0 + blockIdx.x * 256 + t;
  if (row >= v.NP) return;
  double* arow = A + (size_t)row * ld + c0;
  double x[64];
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const v2d w = *reinterpret_cast<const v2d*>(arow + 2 * q);
    x[2 * q] = w[0];
    x[2 * q + 1] = w[1];
  }
  forward_subst_64(Lb, x);
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    v2d w;
    w[0] = x[2 * q];
    w[1] = x[2 * q + 1];
    *reinterpret_cast<v2d*>(arow + 2 * q) = w;
  }
}

// ---------------------------------------------------------------------------------------------
// logdet = 2 sum_{i<n} log L_ii ;  yty = sum_{c<n} L[n,c]^2   (row n of the factor holds y^T)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logdet_kernel(BatchView v, double* __restrict__ logdet, double* __restrict__ yty) {
  __shared__ double red[2][256];
  const int emu = slot_emu(v.idx, blockIdx.x);
  const int ld = v.NP;
  const double* A = v.A + (size_t)emu * ld * ld;
  double s = 0., q = 0.;
  for (int i = threadIdx.x; i < v.n; i += 256) {
    s += log(A[(size_t)i * ld + i]);
    const double y = A[(size_t)v.n * ld + i];
    q += y * y;
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    logdet[emu] = 2.0 * red[0][0];
    yty[emu] = red[1][0];
  }
}

// ---------------------------------------------------------------------------------------------
// alpha = L^-T y, blocked right-looking back substitution, one workgroup per emulator.
//   for kb = last .. 0:  alpha_kb = L_kk^-T w_kb ;  w[0:k0] -= L[kb rows, 0:k0]^T alpha_kb
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void backsolve_kernel(BatchView v) {
  __shared__ double Lb[64][65];
  __shared__ double ab[64];
  const int emu = slot_emu(v.idx, blockIdx.x);
  const int ld = v.NP, n = v.n;
  const double* A = v.A + (size_t)emu * ld * ld;
  double* w = v.alpha + (size_t)emu * ld;
  const int t = threadIdx.x;
  for (int i = t; i < ld; i += 256) w[i] = (i < n) ? A[(size_t)n * ld + i] : 0.0;
  __syncthreads();
  const int nblk = (n + 63) / 64;
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int k0 = kb * 64;
    for (int e = t; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      Lb[r][c] = A[(size_t)(k0 + r) * ld + k0 + c];
    }
    if (t < 64) ab[t] = w[k0 + t];
    __syncthreads();
    // transposed triangular solve inside the block (rows >= n of the block are identity padding,
    // except row n which holds y: it must not take part)
    if (t < 64) {
      for (int j = 63; j >= 0; --j) {
        // lane j finalises, the others subtract
        double aj = ab[j] / Lb[j][j];
        if (k0 + j >= n) aj = 0.0;
        if (t == j) ab[j] = aj;
        if (t < j) ab[t] -= Lb[j][t] * aj;      // L^T[t][j] = L[j][t]
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    if (t < 64) w[k0 + t] = ab[t];
    // w[c] -= sum_r L[k0+r][c] alpha[k0+r], c < k0 : each thread owns columns, rows are contiguous
    for (int c = t; c < k0; c += 256) {
      double s = 0.;
#pragma unroll 8
      for (int r = 0; r < 64; ++r) s = __builtin_fma(A[(size_t)(k0 + r) * ld + c], ab[r], s);
      w[c] -= s;
    }
    __syncthreads();
  }
}

// alpha = Linv^T y  (used when Linv is already available): alpha_i = sum_{k>=i, k<n} Linv[k][i] y_k
__global__ __launch_bounds__(256) void alpha_linv_kernel(BatchView v) {
  const int emu = slot_emu(v.idx, blockIdx.y);
  const int ld = v.NP, n = v.n;
  const double* Li = v.Linv + (size_t)emu * ld * ld;
  const double* y = v.A + (size_t)emu * ld * ld + (size_t)n * ld;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ld) return;
  double s = 0.;
  if (i < n)
    for (int k = i; k < n; ++k) s = __builtin_fma(Li[(size_t)k * ld + i], y[k], s);
  v.alpha[(size_t)emu * ld + i] = s;
}

// ---------------------------------------------------------------------------------------------
// trtri leaf: invert every 64x64 diagonal block of L; lane j produces column j of the inverse.
// Also zeroes the block to the right inside the same 128-tile so that 128-granular consumers can
// treat diagonal tiles of Linv as dense.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void trtri_leaf_kernel(BatchView v) {
  __shared__ __attribute__((aligned(16))) double Lb[64 * 64];
  const int emu = slot_emu(v.idx, blockIdx.y);
  const int ld = v.NP;
  const int d0 = blockIdx.x * 64;
  const double* L = v.A + (size_t)emu * ld * ld;
  double* Li = v.Linv + (size_t)emu * ld * ld;
  const int t = threadIdx.x;
  for (int e = t; e < 64 * 64; e += 64) {
    const int r = e >> 6, c = e & 63;
    Lb[e] = (c <= r) ? L[(size_t)(d0 + r) * ld + d0 + c] : 0.0;
  }
  __syncthreads();
  double x[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) x[i] = (i == t) ? 1.0 : 0.0;
  forward_subst_64(Lb, x);
#pragma unroll
  for (int i = 0; i < 64; ++i) Li[(size_t)(d0 + i) * ld + d0 + t] = x[i];
  if ((blockIdx.x & 1) == 0 && d0 + 64 < ld) {
#pragma unroll 8
    for (int i = 0; i < 64; ++i) Li[(size_t)(d0 + i) * ld + d0 + 64 + t] = 0.0;
  }
}

// out (n,n) <- src (NP,NP).  mode 0: copy; 1: transpose; 2: symmetric from the lower triangle
__global__ void extract_kernel(const double* __restrict__ src, int NP, int n, double* __restrict__ out, int mode) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n) return;
  double x;
  if (mode == 0) x = src[(size_t)i * NP + j];
  else if (mode == 1) x = src[(size_t)j * NP + i];
  else x = (j <= i) ? src[(size_t)i * NP + j] : src[(size_t)j * NP + i];
  out[(size_t)i * n + j] = x;
}

// =============================================================================================
void launch_potf2(const BatchView& v, int c0, int* info, hipStream_t s) {
  hipLaunchKernelGGL(potf2_kernel, dim3(v.nb), dim3(256), 0, s, v, c0, info);
}

void launch_trsm(const BatchView& v, int c0, int r0, hipStream_t s) {
  const int rows = v.NP - r0;
  if (rows <= 0) return;
  hipLaunchKernelGGL(trsm_kernel, dim3((rows + 255) / 256, v.nb), dim3(256), 0, s, v, c0, r0);
}

void launch_logdet(const BatchView& v, double* logdet, double* yty, hipStream_t s) {
  hipLaunchKernelGGL(logdet_kernel, dim3(v.nb), dim3(256), 0, s, v, logdet, yty);
}

void launch_backsolve(const BatchView& v, hipStream_t s) {
  hipLaunchKernelGGL(backsolve_kernel, dim3(v.nb), dim3(256), 0, s, v);
}

void launch_alpha_from_linv(const BatchView& v, hipStream_t s) {
  hipLaunchKernelGGL(alpha_linv_kernel, dim3((v.NP + 255) / 256, v.nb), dim3(256), 0, s, v);
}

void launch_trtri_merges(const BatchView& v, hipStream_t s);   // kernels_gemm.hip

void launch_trtri(const BatchView& v, hipStream_t s) {
  hipLaunchKernelGGL(trtri_leaf_kernel, dim3(v.NP / 64, v.nb), dim3(64), 0, s, v);
  launch_trtri_merges(v, s);
}

void launch_extract(const double* src, int NP, int n, double* out, int mode, hipStream_t s) {
  hipLaunchKernelGGL(extract_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, src, NP, n, out, mode);
}

}  // namespace mogp
