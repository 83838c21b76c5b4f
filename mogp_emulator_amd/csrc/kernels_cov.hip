// Distance + covariance kernels (HBM-bound): K build, cross-covariance with fused predictive
// mean, input-derivative of the mean, and the fused log-posterior-gradient reduction.
//
// They replace the reference's thread-per-entry CUDA kernels and its materialised derivative
// planes:  *_cov_batch_kernel (kernel.cu:55-65, 251-261), *_cov_deriv_x_batch_kernel
// (kernel.cu:69-100, 264-302), *_cov_deriv_theta_batch_kernel + 3 gemv (kernel.cu:106-141,
// 305-348; densegp_gpu.hpp:679-730); arithmetic follows the CPU oracle, Kernel.py:444-485,
// 772-814, 861-906 (sigma^2 multiplies the kernel value outside the exponential,
// GaussianProcess.py:542).
//
// Tiling: a 64x64 output tile per 256-thread workgroup; the two 64-row blocks of X are staged
// once in LDS as [D][64] (coalesced global reads of contiguous row blocks, conflict-free LDS
// reads because consecutive lanes touch consecutive rows); each thread owns a 4x4 micro tile so
// every staged coordinate is reused 4 times from registers and K rows are written as 32-byte
// pieces of full 512-byte row segments.  exp(theta_d) is precomputed once per emulator on the
// host (parameter block P), not per pair as in the reference.
#include <algorithm>
#include <type_traits>
#include "launch.h"
#include "cov_dev.h"

namespace mogp {

typedef double v2d __attribute__((ext_vector_type(2)));

// K build and cross covariance of the squared-exponential / Matern-5/2 kernels on inputs scaled by sqrt(e_d) at staging (micro_r2<.., SC>).
// Round 6, device time unscaled / scaled (profiles/r06_cov_scaled_ab.txt): cross covariance 64 x n=2000 x 10^4 points, d = 10: 2.89 -> 2.45 ms
// (4.2 TB/s of written bytes); C4 (16 x n=5000, Matern-5/2, d = 20) 3.87 -> 3.07 ms; K build C4 0.79 -> 0.75 ms, headline 0.28 -> 0.27 (it
// sits on its write floor); K entries agree to the last digit printed, log-posteriors to 2e-12 (cond 2e9) / 3e-15 (C4).
constexpr bool COV_SCALED = true;

__device__ __forceinline__ int slot_emu2(const int* idx, int z) { return idx ? idx[z] : z; }

// r2 for the thread's RA x CB micro tile (rows RA*ty.., cols CB*tx..): 4 x 4 with (ty, tx) = (t >> 4, t & 15), or 8 x 2 with
// (t >> 5, t & 31) -- the latter makes a wave's store instruction two whole 512-byte rows of the 64-column tile
// SC (round 6): the staged coordinates are already multiplied by sqrt(e_d) (stage_rows<true>): r2 = sum_d (u_id - u_jd)^2, two vector
// instructions per pair and dimension instead of three.  The difference form is kept: r2 of REPEATED points is exactly 0 (identical rows of K:
// what nugget="pivot" and the jitter ladder rest on) and of near-coincident points relatively accurate; the Gram form a_i + a_j - 2 b_ij
// through the matrix cores carries an absolute error ~eps max a_i into every r2 -- harmless where all pairs are far apart
// (the benchmark designs: profiles/r06_gram_form_accuracy.txt), but it gives repeated points r2 = +-1e-15.
template <int RA, int CB, bool SC = false>
__device__ __forceinline__ void micro_r2(const double* si, const double* sj, const double* __restrict__ P, int D, int ty, int tx,
                                         double (&r2)[RA][CB]) {
#pragma unroll
  for (int a = 0; a < RA; ++a)
#pragma unroll
    for (int b = 0; b < CB; ++b) r2[a][b] = 0.0;
  for (int d = 0; d < D; ++d) {
    const double e = P[d];
    double xi[RA], xj[CB];
#pragma unroll
    for (int a = 0; a < RA; ++a) xi[a] = si[d * 64 + RA * ty + a];
#pragma unroll
    for (int b = 0; b < CB; ++b) xj[b] = sj[d * 64 + CB * tx + b];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
      for (int b = 0; b < CB; ++b) {
        const double df = xi[a] - xj[b];
        r2[a][b] = SC ? __builtin_fma(df, df, r2[a][b]) : __builtin_fma(e * df, df, r2[a][b]);
      }
  }
}

// Product-of-Matern-5/2 kernel (KT = 2, Kernel.py:581-763, 986-997): k = prod_d m52(e_d (x_d - y_d)^2).
// (dm52/dr2) / m52 needs no exponential: -(5/6)(1+s) / (1+s+s^2/3), s = sqrt(5 r2).
__device__ __forceinline__ double mat52_dlog(double r2) {
  const double s = sqrt(5.0 * r2);
  return -(5.0 / 6.0) * (1.0 + s) / (1.0 + s + (5.0 / 3.0) * r2);
}

// Orders the work on a register tile in groups of four entries: group g's results and group g+1's inputs pass through ONE empty asm
// statement, so nothing of group g+1 can start before group g is finished.  Left to itself the compiler interleaves all sixteen
// exponentials (about ten live registers each: 107 instead of 60 VGPRs for the K build, half the waves per SIMD -- and these kernels
// live on occupancy).  Entry e of the tile is t[e / CB][e % CB].
template <int RA, int CB>
__device__ __forceinline__ void group_fence(double (&t)[RA][CB], int g) {
  static_assert(RA * CB == 16, "sixteen entries per thread");
  double* f = &t[0][0];
  if (g < 3)
    asm volatile("" : "+v"(f[4 * g]), "+v"(f[4 * g + 1]), "+v"(f[4 * g + 2]), "+v"(f[4 * g + 3]), "+v"(f[4 * g + 4]), "+v"(f[4 * g + 5]),
                 "+v"(f[4 * g + 6]), "+v"(f[4 * g + 7]));
}

// kernel values (without sigma^2) of the thread's micro tile
template <int KT, int RA, int CB, bool SC = false>
__device__ __forceinline__ void micro_k(const double* si, const double* sj, const double* __restrict__ P, int D, int ty, int tx,
                                        double (&k)[RA][CB], const double* etab) {
  double* kf = &k[0][0];
  if (KT < 2) {
    micro_r2<RA, CB, SC>(si, sj, P, D, ty, tx, k);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int e = 4 * g; e < 4 * g + 4; ++e) kf[e] = kern_val<KT>(kf[e], etab);
      group_fence<RA, CB>(k, g);
    }
    return;
  }
  // prod_d (1 + s_d + s_d^2/3) * exp(-sum_d s_d): one exponential per pair
  double ssum[RA][CB];
#pragma unroll
  for (int a = 0; a < RA; ++a)
#pragma unroll
    for (int b = 0; b < CB; ++b) {
      k[a][b] = 1.0;
      ssum[a][b] = 0.0;
    }
  for (int d = 0; d < D; ++d) {
    const double e = P[d];
    double xi[RA], xj[CB];
#pragma unroll
    for (int a = 0; a < RA; ++a) xi[a] = si[d * 64 + RA * ty + a];
#pragma unroll
    for (int b = 0; b < CB; ++b) xj[b] = sj[d * 64 + CB * tx + b];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
      for (int b = 0; b < CB; ++b) {
        const double df = xi[a] - xj[b];
        const double r2 = e * df * df;
        const double sd = sqrt(5.0 * r2);
        k[a][b] *= 1.0 + sd + (5.0 / 3.0) * r2;
        ssum[a][b] += sd;
      }
  }
  double* sf = &ssum[0][0];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int e = 4 * g; e < 4 * g + 4; ++e) kf[e] *= lean_exp_neg<false>(sf[e], etab);
    group_fence<RA, CB>(ssum, g);
  }
}

// ---------------------------------------------------------------------------------------------
// K build into the factor buffer A (lower tiles only, diagonal tiles in full), with the nugget
// fused on the diagonal, the targets laid into row n and identity padding beyond.
// ---------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void cov_build_kernel(BatchView v, ZeroRanges zero) {
  if (blockIdx.y == 0 && blockIdx.x < 2 && zero.p[blockIdx.x])          // (workgroups (0, 0) and (1, 0): one range each; NP >= 128 gives three tiles)
    for (unsigned e = threadIdx.x; e < zero.n[blockIdx.x]; e += 256) zero.p[blockIdx.x][e] = 0u;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ double etab[256];
  stage_exp_tab(etab);
  const int z = blockIdx.y;
  const int emu = slot_emu2(v.idx, z);
  const int tile = blockIdx.x;
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tile) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const int i0 = ti * 64, j0 = tj * 64;
  const int n = v.n, D = v.D, ld = v.LD;
  const double* P = v.P + (size_t)emu * v.PS;
  double* A = v.A + (size_t)emu * v.MS;
  const double* T = v.T + (size_t)emu * n;
  double* si = sm;
  double* sj = sm + 64 * D;
  constexpr bool SC = COV_SCALED && KT < 2;
  stage_rows<SC>(v.X + (size_t)emu * v.XS, n, D, i0, si, P);
  stage_rows<SC>(v.X + (size_t)emu * v.XS, n, D, j0, sj, P);
  __syncthreads();
  // 8 x 2 micro tile: thread (ty, tx) = (t >> 5, t & 31) owns rows 8 ty .. 8 ty + 7 and columns 2 tx, 2 tx + 1, so that one store
  // instruction of a wave is two whole 512-byte rows of the tile (with 4 x 4 micro tiles it was sixteen 16-byte pieces per row pair
  // at a 32-byte stride: every 128-byte line written by two instructions, half each)
  const int ty = threadIdx.x >> 5, tx = threadIdx.x & 31;
  double kv[8][2];
  micro_k<KT, 8, 2, SC>(si, sj, P, D, ty, tx, kv, etab);
  const double sig2 = P[D], nug = P[D + 1];
  // INTERIOR: a tile strictly below the diagonal whose rows are all training points (15 of 16 tiles at n = 2000): every entry is
  // sigma^2 k -- no nugget, no target row, no padding
  const bool interior = ti > tj && i0 + 64 <= n;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int i = i0 + 8 * ty + a;
    double out[2];
    if (interior) {
#pragma unroll
      for (int b = 0; b < 2; ++b) out[b] = sig2 * kv[a][b];
    } else {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = j0 + 2 * tx + b;
        out[b] = cov_entry(v, T, i, j, sig2 * kv[a][b], nug);
      }
    }
    *reinterpret_cast<double2*>(A + (size_t)i * ld + j0 + 2 * tx) = make_double2(out[0], out[1]);
  }
  // single right-hand side: the solution vector starts as all-ones bit patterns -- the "not there yet" value the chunks of the
  // one-launch back substitution poll for (backsolve_chain_kernel<SENT>, kernels_chol.hip); every other solve overwrites it
  if (ti == tj && v.R == 1 && v.Z && threadIdx.x < 64)
    reinterpret_cast<unsigned long long*>(v.Z + (size_t)emu * ld)[i0 + threadIdx.x] = ~0ull;
}

// full (n,n) sigma^2 k(X,X) without nugget for get_K (GaussianProcessGPU.py:504-513)
template <int KT>
__global__ __launch_bounds__(256) void cov_full_kernel(BatchView v, int emu, const double* __restrict__ Xp, int n, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ double etab[256];
  stage_exp_tab(etab);
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int D = v.D;
  if (emu < 0) {                     // batched form: one (n, n) matrix per slot
    emu = slot_emu2(v.idx, blockIdx.z);
    out += (size_t)blockIdx.z * n * n;
  }
  const double* P = v.P + (size_t)emu * v.PS;
  double* si = sm;
  double* sj = sm + 64 * D;
  stage_rows(Xp, n, D, i0, si);
  stage_rows(Xp, n, D, j0, sj);
  __syncthreads();
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  double kv[4][4];
  micro_k<KT, 4, 4>(si, sj, P, D, ty, tx, kv, etab);
  const double sig2 = P[D];
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) {
      const int i = i0 + 4 * ty + a, j = j0 + 4 * tx + b;
      if (i < n && j < n) out[(size_t)i * n + j] = sig2 * kv[a][b];
    }
}

// ---------------------------------------------------------------------------------------------
// Cross covariance Ks[z][m][j] = sigma^2 k(x*_m, x_j) (zero for j >= n, m >= mtot) with the
// predictive mean fused: mean[z][m] = sum_j Ks[m][j] alpha_j.  One workgroup per 64 test points,
// sweeping all training-point tiles, so the mean needs no atomics and is deterministic.
// ---------------------------------------------------------------------------------------------
// RB = compile-time bound of the right-hand-side rows: 1 for the plain path (8 instead of 64 accumulator registers,
// five instead of three waves per SIMD), RMAX with an analytic mean.
// NPF > 0 (round 5; D <= 4 NPF): the NEXT tile of training inputs (and of alpha) is requested into NPF registers per thread before the
// current tile is computed and deposited into the other of two LDS buffers after it -- one barrier per tile, no exposed load.  Before,
// every one of the NP / 64 tiles of a workgroup began with a global load between two barriers: 10 us per tile of which 1.4 us per wave
// is arithmetic; with four waves per SIMD that is what bounded the kernel (3.2 TB/s of writes whatever the store pattern or the
// instruction count).  NPF = 0: the synchronous form (D > 32: the register set would be larger than what it hides).
template <int KT, int RB, int NPF>
__global__ __launch_bounds__(256) void cross_cov_mean_kernel(BatchView v, const double* __restrict__ Xs, int m, int MP,
                                                           double* __restrict__ Ks, double* __restrict__ mean, int mean_ld) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ double etab[256];
  stage_exp_tab(etab);
  const int z = blockIdx.y;
  const int emu = slot_emu2(v.idx, z);
  const int n = v.n, D = v.D, ld = v.LD, R = (RB == 1) ? 1 : v.R;
  const int i0 = blockIdx.x * 64;
  const double* P = v.P + (size_t)emu * v.PS;
  const double* alpha0 = v.alpha + (size_t)emu * v.RA * ld;   // row 0: K^-1 (t - H beta)
  const double* Zr = v.Z + (size_t)emu * R * ld;              // rows 1..: K^-1 h_c
  const double* Xtr = v.X + (size_t)emu * v.XS;
  constexpr int NB = NPF > 0 ? 2 : 1;                         // LDS buffers of the training tile / the alpha tile
  constexpr int NPA = RB == 1 ? 1 : (RMAX * 64 + 255) / 256;  // alpha values per thread
  double* si = sm;
  double* sj = sm + 64 * D;           // NB x [D][64]
  double* sa = sj + NB * 64 * D;      // NB x [R][64] operand tile
  double* red = sa + NB * RMAX * 64;  // 64 x 33
  constexpr bool SC = COV_SCALED && KT < 2;
  stage_rows<SC>(Xs, m, D, i0, si, P);
  // 8 x 2 micro tile (see cov_build_kernel): a wave's store instruction is two whole 512-byte rows of the K* tile
  const int t = threadIdx.x, ty = t >> 5, tx = t & 31;
  const double sig2 = P[D];
  double macc[RB][8];
#pragma unroll
  for (int c = 0; c < RB; ++c)
#pragma unroll
    for (int a = 0; a < 8; ++a) macc[c][a] = 0.;
  double* Kz = Ks ? Ks + (size_t)z * MP * ld : nullptr;
  const int ntj = v.NP / 64;
  // prefetch registers and the (tile-independent) LDS slot of each: element e = t + 256 q of a [64][D] row block goes to [d][row]
  double px[NPF > 0 ? NPF : 1], pa[NPA];
  double psc[NPF > 0 ? NPF : 1];                              // SC: sqrt(e_d) of the coordinate a prefetch register carries
  int loff[NPF > 0 ? NPF : 1];
  const int cnt = 64 * D;
  if (NPF > 0) {
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
      const int e = t + 256 * q, r = e / D;
      loff[q] = (e - r * D) * 64 + r;
      psc[q] = (SC && e < cnt) ? sqrt(P[e - r * D]) : 1.0;
    }
  }
  auto fetch = [&](int j0) {
    const int avail = max(0, min(64, n - j0)) * D;
    const double* src = Xtr + (size_t)j0 * D;
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
      const int e = t + 256 * q;
      px[q] = (e < avail) ? src[e] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      const int e = t + 256 * q, c = e >> 6, jj = e & 63;
      const double* srca = (c == 0) ? alpha0 : Zr + (size_t)c * ld;
      pa[q] = (e < R * 64 && j0 + jj < n) ? srca[j0 + jj] : 0.0;
    }
  };
  auto deposit = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NPF; ++q)
      if (t + 256 * q < cnt) sj[buf * cnt + loff[q]] = SC ? px[q] * psc[q] : px[q];
#pragma unroll
    for (int q = 0; q < NPA; ++q)
      if (t + 256 * q < R * 64) sa[buf * RMAX * 64 + t + 256 * q] = pa[q];
  };
  if (NPF > 0) {
    fetch(0);
    deposit(0);
  }
  for (int tj = 0; tj < ntj; ++tj) {
    const int j0 = tj * 64;
    const int cur = NPF > 0 ? (tj & 1) : 0;
    const double* sjc = sj + cur * cnt;
    const double* sac = sa + cur * RMAX * 64;
    __syncthreads();
    if (NPF == 0) {
      if (j0 < n) {
        stage_rows<SC>(Xtr, n, D, j0, sj, P);
        for (int e = t; e < R * 64; e += 256) {
          const int c = e >> 6, jj = e & 63;
          const double* src = (c == 0) ? alpha0 : Zr + (size_t)c * ld;
          sa[e] = (j0 + jj < n) ? src[j0 + jj] : 0.0;
        }
      }
      __syncthreads();
    } else if (j0 + 64 < n) {
      fetch(j0 + 64);
    }
    double kv[8][2];
    if (j0 < n) micro_k<KT, 8, 2, SC>(si, sjc, P, D, ty, tx, kv, etab);
    double* Kt = Kz + (size_t)(i0 + 8 * ty) * ld + j0 + 2 * tx;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int i = i0 + 8 * ty + a;
      double out[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = j0 + 2 * tx + b;
        double x = 0.0;
        if (j < n && i < m) x = sig2 * kv[a][b];
        out[b] = x;
        if (j0 < n) {
#pragma unroll
          for (int c = 0; c < RB; ++c)
            if (c < R) macc[c][a] = __builtin_fma(x, sac[c * 64 + 2 * tx + b], macc[c][a]);
        }
      }
      if (Kz) *reinterpret_cast<double2*>(Kt + (size_t)a * ld) = make_double2(out[0], out[1]);
    }
    // (the other buffer was last read in the previous iteration, which every wave left through the barrier above)
    if (NPF > 0 && j0 + 64 < n) deposit(cur ^ 1);
  }
#pragma unroll
  for (int c = 0; c < RB; ++c) {
    if (c < R) {
      __syncthreads();
#pragma unroll
      for (int a = 0; a < 8; ++a) red[(8 * ty + a) * 33 + tx] = macc[c][a];
      __syncthreads();
      if (t < 64) {
        double s = 0.;
        for (int q = 0; q < 32; ++q) s += red[t * 33 + q];
        const int i = i0 + t;
        if (i < m) mean[((size_t)z * R + c) * mean_ld + i] = s;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// d mean / d x*:  deriv[z][m][d] = sum_j sigma^2 dk/dr2(r2_mj) * 2 e_d (x*_md - x_jd) alpha_j
// ---------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void predict_deriv_kernel(BatchView v, const double* __restrict__ Xs, int m,
                                                          double* __restrict__ deriv, long deriv_stride) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ double etab[256];
  stage_exp_tab(etab);
  const int z = blockIdx.y;
  const int emu = slot_emu2(v.idx, z);
  const int n = v.n, D = v.D, ld = v.LD;
  const int i0 = blockIdx.x * 64;
  const double* P = v.P + (size_t)emu * v.PS;
  const double* alpha = v.alpha + (size_t)emu * v.RA * ld;
  double* si = sm;                    // [D][64] test points
  double* sj = sm + 64 * D;           // [D][64] training points
  double* G = sm + 128 * D;           // [64][65]  G[m][j] = sig2 * dk/dr2 * 2 * alpha_j
  double* dacc = G + 64 * 65;         // [64][D]
  stage_rows(Xs, m, D, i0, si);
  for (int e = threadIdx.x; e < 64 * D; e += 256) dacc[e] = 0.0;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const double sig2 = P[D];
  const int row = threadIdx.x >> 2, part = threadIdx.x & 3;     // phase 2 mapping
  for (int j0 = 0; j0 < n; j0 += 64) {
    __syncthreads();
    stage_rows(v.X + (size_t)emu * v.XS, n, D, j0, sj);
    __syncthreads();
    // KT < 2: G = 2 sigma^2 dk/dr2 alpha_j;  product kernel: G = 2 sigma^2 k alpha_j and the per-dimension
    // factor (dm52/dr2)/m52 of dimension d is applied in the contraction below
    double r2[4][4];
    if (KT < 2) micro_r2<4, 4>(si, sj, P, D, ty, tx, r2);
    else micro_k<KT, 4, 4>(si, sj, P, D, ty, tx, r2, etab);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = j0 + 4 * tx + b;
        const double f = (KT < 2) ? kern_dr2<KT>(r2[a][b], etab) : r2[a][b];
        G[(4 * ty + a) * 65 + 4 * tx + b] = (j < n) ? 2.0 * sig2 * f * alpha[j] : 0.0;
      }
    __syncthreads();
    if (KT < 2) {
      // sum_j G_mj (x*_md - x_jd) = x*_md sum_j G_mj - sum_j G_mj x_jd: the thread's 16 values of G stay in registers for all D
      // dimensions and every pair costs ONE fused multiply-add per dimension (before: G re-read from LDS for every dimension,
      // two LDS reads and two operations per pair and dimension -- the kernel was LDS-bound, 10 ms for 64 x 10^4 points x n=2000).
      // The coordinates are taken relative to the workgroup's first test point, so that the two sums stay of the size of the
      // differences.
      double g[16], gsum = 0.;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        g[q] = G[row * 65 + part * 16 + q];
        gsum += g[q];
      }
      for (int d = 0; d < D; ++d) {
        const double x0 = si[d * 64];
        const double* xj = sj + d * 64 + part * 16;
        double u = 0.;
#pragma unroll
        for (int q = 0; q < 16; ++q) u = __builtin_fma(g[q], xj[q] - x0, u);
        double s = __builtin_fma(si[d * 64 + row] - x0, gsum, -u);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (part == 0) dacc[row * D + d] += P[d] * s;
      }
    } else {
      for (int d = 0; d < D; ++d) {
        const double xm = si[d * 64 + row];
        double s = 0.;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
          const int j = part * 16 + q;
          const double df = xm - sj[d * 64 + j];
          s = __builtin_fma(G[row * 65 + j] * mat52_dlog(P[d] * df * df), df, s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (part == 0) dacc[row * D + d] += P[d] * s;
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * D; e += 256) {
    const int r = e / D;
    if (i0 + r < m) deriv[(size_t)z * deriv_stride + (size_t)(i0 + r) * D + (e - r * D)] = dacc[e];
  }
}

// ---------------------------------------------------------------------------------------------
// Fused gradient reduction.  For every lower tile of W = Kinv - alpha alpha^T:
//   s_p   = sum_ij w_ij W_ij sigma^2 dk/dr2(r2_ij) e_p (x_ip - x_jp)^2     p < D
//   s_D   = sum_ij w_ij W_ij K_ij                                           (covariance scale)
//   s_D+1 = sum_i Kinv_ii ,  s_D+2 = sum_i alpha_i^2                         (nugget)
// with w_ij = 2 below the diagonal, 1 on it.  K and dK/dtheta are recomputed from LDS-staged X,
// never materialised (the reference writes and re-reads (D+1) n x n planes).
// partial[(z*ntiles + tile)*(D+3) + p]
// ---------------------------------------------------------------------------------------------
// three waves per SIMD (<= 168 VGPRs; the squared-exponential instantiation spills five dwords for it) -- the product kernel needs 255
template <int KT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KT == 2 ? 2 : 3))) void grad_kernel(BatchView v, int ntiles, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ double etab[256];
  stage_exp_tab(etab);
  const int z = blockIdx.y;
  const int emu = slot_emu2(v.idx, z);
  int tile = blockIdx.x;
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tile) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const int i0 = ti * 64, j0 = tj * 64;
  const int n = v.n, D = v.D, ld = v.LD;
  const double* P = v.P + (size_t)emu * v.PS;
  const double* Ki = v.Kinv + (size_t)emu * v.MS;
  // W = Kinv - sum_c g_c g_c^T over the gradient rows of alpha: the single row when R = 1, rows 1..R otherwise
  const int R = v.RA - (v.R > 1 ? 1 : 0);
  const double* alpha = v.alpha + ((size_t)emu * v.RA + (v.R > 1 ? 1 : 0)) * ld;
  double* si = sm;
  double* sj = sm + 64 * D;
  double* red = sm + 128 * D;         // [8 quantities][256 threads]
  // the thread's 4 x 4 entries of K^-1, requested before anything else as 16-byte pieces: the buffer holds NP x LD entries, so
  // every address of a tile is valid whatever n is (entries that do not count are masked below)
  v2d kin2[4][2];
  {
    const double* kp = Ki + (size_t)(i0 + 4 * (threadIdx.x >> 4)) * ld + j0 + 4 * (threadIdx.x & 15);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      kin2[a][0] = *reinterpret_cast<const v2d*>(kp + (size_t)a * ld);
      kin2[a][1] = *reinterpret_cast<const v2d*>(kp + (size_t)a * ld + 2);
    }
  }
  stage_rows(v.X + (size_t)emu * v.XS, n, D, i0, si);
  stage_rows(v.X + (size_t)emu * v.XS, n, D, j0, sj);
  __syncthreads();
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  // KT < 2: r2 holds squared distances; product kernel: r2 holds the kernel values themselves
  double r2[4][4];
  if (KT < 2) micro_r2<4, 4>(si, sj, P, D, ty, tx, r2);
  else micro_k<KT, 4, 4>(si, sj, P, D, ty, tx, r2, etab);
  const double sig2 = P[D];
  double scov = 0., strace = 0., saa = 0.;
  // W = K^-1 - (rank-R correction sum_c g_c[i] g_c[j]; R = 1: alpha_i alpha_j) for the micro tile
  double W[4][4], gsq[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    gsq[a] = 0.;
#pragma unroll
    for (int b = 0; b < 4; ++b) W[a][b] = 0.;
  }
  for (int c = 0; c < R; ++c) {
    const double* g = alpha + (size_t)c * ld;
    double gi[4], gj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) gi[a] = (i0 + 4 * ty + a < n) ? g[i0 + 4 * ty + a] : 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) gj[b] = (j0 + 4 * tx + b < n) ? g[j0 + 4 * tx + b] : 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      gsq[a] = __builtin_fma(gi[a], gi[a], gsq[a]);
#pragma unroll
      for (int b = 0; b < 4; ++b) W[a][b] = __builtin_fma(-gi[a], gj[b], W[a][b]);
    }
  }
  // Branch-free over the 16 pairs: with `if (weight != 0)` around each pair every K^-1 entry was one 8-byte load inside its own
  // basic block -- 16 dependent load -> exp -> next-branch round trips per thread, and the constants of the exponential were
  // materialised 16 times (78 instructions per pair; 0.63 ms for 64 x n=2000, 1.8 x the kernel's vector-ALU floor).
  double G[4][4];                     // w * W * sigma^2 * dk/dr2   (product kernel: w * W * sigma^2 * k)
  // INTERIOR: a tile strictly below the diagonal whose rows are all real -- every pair counts twice, nothing to mask (15 of 16
  // tiles at n = 2000)
  auto pairs = [&](auto interior_) {
    constexpr bool INTERIOR = decltype(interior_)::value;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int i = i0 + 4 * ty + a;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = j0 + 4 * tx + b;
        const double kin = (b < 2) ? kin2[a][0][b] : kin2[a][1][b - 2];
        // weight of the pair in the symmetric sums: 2 strictly below the diagonal, 1 on it, 0 above it and in the padding
        const bool live = INTERIOR || ((i < n) && (j <= i));
        const double w = INTERIOR ? 2.0 : (live ? ((j < i) ? 2.0 : 1.0) : 0.0);
        const double Wv = live ? kin + W[a][b] : 0.0;
        const double kval = (KT < 2) ? kern_val<KT>(r2[a][b], etab) : r2[a][b];
        const double wk = w * Wv * sig2;
        scov = __builtin_fma(wk, kval, scov);
        G[a][b] = (KT < 2) ? wk * kern_dr2<KT>(r2[a][b], etab) : wk * kval;
        if (!INTERIOR && live && i == j) {
          strace += kin;
          saa += gsq[a];
        }
      }
    }
  };
  if (ti > tj && i0 + 64 <= n) pairs(std::true_type());
  else pairs(std::false_type());
  // Workgroup sums of the D + 3 quantities, eight at a time through LDS: thread t parks its partial of quantity q in
  // red[q & 7][t]; then thread (q', part) = (t >> 5, t & 31) adds eight neighbours and the 32 parts meet in five shuffles -- one
  // dependent chain per eight quantities (a wave reduction per quantity was a chain of six LDS-crossbar round trips each).
  const int NQ = D + 3;
  const int t = threadIdx.x;
  auto flush = [&](int q0) {
    __syncthreads();
    const int q = t >> 5, part = t & 31;
    const v2d* rp = reinterpret_cast<const v2d*>(red + q * 256 + part * 8);
    const v2d u0 = rp[0], u1 = rp[1], u2 = rp[2], u3 = rp[3];
    double s = ((u0[0] + u0[1]) + (u1[0] + u1[1])) + ((u2[0] + u2[1]) + (u3[0] + u3[1]));
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (part == 0 && q0 + q < NQ) partial[((size_t)z * ntiles + tile) * NQ + q0 + q] = s;
    __syncthreads();
  };
  for (int p = 0; p < NQ; ++p) {
    double s;
    if (p < D) {
      s = 0.;
      double xi[4], xj[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) xi[a] = si[p * 64 + 4 * ty + a];
#pragma unroll
      for (int b = 0; b < 4; ++b) xj[b] = sj[p * 64 + 4 * tx + b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double df = xi[a] - xj[b];
          // d k / d theta_p = dk/dr2 * e_p df^2 ; product kernel: k * (dm52/dr2)/m52 (r2_p) * r2_p,  r2_p = e_p df^2
          if (KT < 2) s = __builtin_fma(G[a][b], df * df, s);
          else s = __builtin_fma(G[a][b] * mat52_dlog(P[p] * df * df), df * df, s);
        }
      s *= P[p];
    } else if (p == D) s = scov;
    else if (p == D + 1) s = strace;
    else s = saa;
    red[(p & 7) * 256 + t] = s;
    if ((p & 7) == 7 || p == NQ - 1) flush(p & ~7);
  }
}

// out[emu][p] = 0.5 * sum_tiles partial  (p <= D);  raw sums for p = D+1, D+2
__global__ __launch_bounds__(256) void grad_finish_kernel(BatchView v, int ntiles, const double* __restrict__ partial, double* __restrict__ out) {
  __shared__ double red[256];
  const int z = blockIdx.y, p = blockIdx.x;
  const int emu = slot_emu2(v.idx, z);
  const int NQ = v.D + 3;
  double s = 0.;
  for (int t = threadIdx.x; t < ntiles; t += 256) s += partial[((size_t)z * ntiles + t) * NQ + p];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(size_t)emu * NQ + p] = (p <= v.D) ? 0.5 * red[0] : red[0];
}

// ---------------------------------------------------------------------------------------------
// Stand-alone kernel objects of the native module (bindings.cu:340-361; kernel.hpp:47-107): covariance of two point
// sets and its derivatives into caller buffers.  mode 0: out[i*n2 + j] = sigma^2 k(x1_i, x2_j) (cov_batch, kernel.cu:55-65);
// mode 1: out[(p*n1 + i)*n2 + j] = d/d theta_p (the CPU class's (n_params, n1, n2) order, Kernel.py:133-173; the
// reference's CUDA kernel indexes its inputs crosswise, kernel.cu:129-141, which is only meaningful for n1 == n2);
// mode 2: out[(j*n1 + i)*D + d] = d/d x1_i[d] (cov_deriv_x_batch, kernel.cu:86-100).  One thread per pair; P = [exp(theta_d),
// sigma^2].  Same per-pair arithmetic as the tiled kernels (pair loop of cov_dev.h), not a hot path.
// ---------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void kernel_object_kernel(const double* __restrict__ x1, int n1, const double* __restrict__ x2, int n2, int D,
                                                            const double* __restrict__ P, int mode, double* __restrict__ out) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)n1 * n2) return;
  const int i = (int)(e / n2), j = (int)(e % n2);
  const double* a = x1 + (size_t)i * D;
  const double* b = x2 + (size_t)j * D;
  const double sig2 = P[D];
  if (KT < 2) {
    double r2 = 0.0;
    for (int d = 0; d < D; ++d) {
      const double df = a[d] - b[d];
      r2 = __builtin_fma(P[d] * df, df, r2);
    }
    const double k = sig2 * kern_val<KT>(r2, EXP_TAB_G), dk = sig2 * kern_dr2<KT>(r2, EXP_TAB_G);
    if (mode == 0) out[e] = k;
    else if (mode == 1) {
      for (int p = 0; p < D; ++p) {
        const double df = a[p] - b[p];
        out[((size_t)p * n1 + i) * n2 + j] = dk * P[p] * df * df;          // dr2/dtheta_p = exp(theta_p) (x - y)_p^2
      }
      out[((size_t)D * n1 + i) * n2 + j] = k;
    } else {
      for (int d = 0; d < D; ++d) out[((size_t)j * n1 + i) * D + d] = dk * 2.0 * P[d] * (a[d] - b[d]);
    }
    return;
  }
  // product of one-dimensional Matern-5/2 factors m(r2_d) = (1 + s + s^2/3) exp(-s), s = sqrt(5 r2_d)
  double k = sig2;
  for (int d = 0; d < D; ++d) {
    const double df = a[d] - b[d];
    k *= kern_val<1>(P[d] * df * df, EXP_TAB_G);
  }
  if (mode == 0) {
    out[e] = k;
    return;
  }
  for (int d = 0; d < D; ++d) {
    const double df = a[d] - b[d], r2 = P[d] * df * df;
    // (dm/dr2) / m = -(5/6) (1 + s) / (1 + s + s^2/3): the exponentials cancel, so nothing is divided by a factor that
    // has underflowed to 0 at large distances (kern_dr2 / kern_val would give 0/0 there)
    const double sd = sqrt(5.0 * r2);
    const double ratio = -(5.0 / 6.0) * (1.0 + sd) / (1.0 + sd + (5.0 / 3.0) * r2);
    if (mode == 1) out[((size_t)d * n1 + i) * n2 + j] = k * ratio * r2;
    else out[((size_t)j * n1 + i) * D + d] = k * ratio * 2.0 * P[d] * df;
  }
  if (mode == 1) out[((size_t)D * n1 + i) * n2 + j] = k;
}

// =============================================================================================
#define KT_DISPATCH(kt, CALL)            \
  do {                                   \
    if ((kt) == 0) { CALL(0); } else if ((kt) == 1) { CALL(1); } else { CALL(2); } \
  } while (0)

void launch_cov_build(const BatchView& v, hipStream_t s, const ZeroRanges& zero) {
  const int nt = v.NP / 64;
  const int ntiles = nt * (nt + 1) / 2;
  const size_t sm = (size_t)128 * v.D * sizeof(double);
  prof_begin("cov_build", s);
#define CALL(K) hipLaunchKernelGGL((cov_build_kernel<K>), dim3(ntiles, v.nb), dim3(256), sm, s, v, zero)
  KT_DISPATCH(v.kernel_type, CALL);
#undef CALL
  // algorithmic bytes: lower triangle written once (4 n^2) + X read once per emulator
  prof_end("cov_build", s, 0., (double)v.nb * (4.0 * (double)v.n * (double)v.n + 8.0 * v.n * v.D));      // algorithmic: the lower triangle of the n x n matrix, not the padded tiles
}

void launch_kernel_object(int kt, const double* x1, int n1, const double* x2, int n2, int D, const double* P, int mode, double* out, hipStream_t s) {
  const dim3 grid((unsigned)(((long)n1 * n2 + 255) / 256));
#define CALL(K) hipLaunchKernelGGL((kernel_object_kernel<K>), grid, dim3(256), 0, s, x1, n1, x2, n2, D, P, mode, out)
  KT_DISPATCH(kt, CALL);
#undef CALL
}

// prior covariance of the test points for every slot: out (nb, m, m) = sigma^2 k(Xs, Xs)
void launch_cov_self_batch(const BatchView& v, const double* Xs, int m, double* out, hipStream_t s) {
  const int nt = (m + 63) / 64;
  const size_t sm = (size_t)128 * v.D * sizeof(double);
#define CALL(K) hipLaunchKernelGGL((cov_full_kernel<K>), dim3(nt, nt, v.nb), dim3(256), sm, s, v, -1, Xs, m, out)
  KT_DISPATCH(v.kernel_type, CALL);
#undef CALL
}

void launch_cov_full(const BatchView& v, int emu, double* out, hipStream_t s) {
  const int nt = (v.n + 63) / 64;
  const size_t sm = (size_t)128 * v.D * sizeof(double);
#define CALL(K) hipLaunchKernelGGL((cov_full_kernel<K>), dim3(nt, nt), dim3(256), sm, s, v, emu, v.X, v.n, out)
  KT_DISPATCH(v.kernel_type, CALL);
#undef CALL
}

void launch_cross_cov_mean(const BatchView& v, const double* Xs, int m, int MP, double* Ks, double* mean, int mean_ld, hipStream_t s) {
  // prefetch registers per thread for the next training tile: 64 D values over 256 threads (0 = the synchronous form)
  const int npf = v.D <= 16 ? 4 : (v.D <= 32 ? 8 : 0);
  const int nbuf = npf ? 2 : 1;
  const size_t sm = (size_t)(64 * v.D + nbuf * (64 * v.D + RMAX * 64) + 64 * 33) * sizeof(double);
  prof_begin("cross_cov", s);
#define LAUNCH(K, RBV, NPFV) \
  hipLaunchKernelGGL((cross_cov_mean_kernel<K, RBV, NPFV>), dim3(MP / 64, v.nb), dim3(256), sm, s, v, Xs, m, MP, Ks, mean, mean_ld)
#define CALL(K)                                                        \
  do {                                                                 \
    if (v.R == 1) {                                                    \
      if (npf == 4) LAUNCH(K, 1, 4);                                   \
      else if (npf == 8) LAUNCH(K, 1, 8);                              \
      else LAUNCH(K, 1, 0);                                            \
    } else {                                                           \
      if (npf == 4) LAUNCH(K, RMAX, 4);                                \
      else if (npf == 8) LAUNCH(K, RMAX, 8);                           \
      else LAUNCH(K, RMAX, 0);                                         \
    }                                                                  \
  } while (0)
  KT_DISPATCH(v.kernel_type, CALL);
#undef CALL
#undef LAUNCH
  prof_end("cross_cov", s, 0., (double)v.nb * (8.0 * (double)m * (double)v.n + 8.0 * ((double)m + v.n) * v.D));      // algorithmic m x n entries, not the padded MP x NP
}

void launch_predict_deriv(const BatchView& v, const double* Xs, int m, double* deriv, long deriv_stride, hipStream_t s) {
  const size_t sm = (size_t)(128 * v.D + 64 * 65 + 64 * v.D) * sizeof(double);
  // (round 5: the next-tile prefetch of the cross covariance was built here too and changed nothing -- 6.2 ms per 64 x 10^4 points either
  // way: this kernel is bound by its arithmetic and the G tile's trip through LDS, not by the staging loads)
  prof_begin("predict_deriv", s);
#define CALL(K) hipLaunchKernelGGL((predict_deriv_kernel<K>), dim3((m + 63) / 64, v.nb), dim3(256), sm, s, v, Xs, m, deriv, deriv_stride)
  KT_DISPATCH(v.kernel_type, CALL);
#undef CALL
  prof_end("predict_deriv", s, 0., (double)v.nb * 8.0 * (double)m * v.D);      // algorithmic bytes: the (m, D) derivatives written
}

int grad_num_tiles(int n) {
  const int nt = (n + 63) / 64;
  return nt * (nt + 1) / 2;
}

// ---------------------------------------------------------------------------------------------
// Gradient terms of the rows of L^-1 that belong to SKIPPED pivots (nugget="pivot" with repeated design points).
// Those rows w are ~ (e_i' - e_i) / d with a replacement diagonal d ~ 1e-6: their contribution w w^T to K^-1 has entries
// ~ 1/d^2 that cancel against identical entries of dK only AFTER the multiplication, which costs ~ eps / d^2 ~ 1e-3 of
// absolute accuracy in sum_ij (K^-1)_ij dK_ij.  So K^-1 is formed without these rows and their share is added as
//      sum_k  w_k^T dK_p w_k ,   u = dK_p w_k first:
// inside u the identical entries of dK meet the +-1/d pair of w and cancel before anything is amplified (the order in
// which the reference's cho_solve-based logdet_deriv, linalg_utils.py:170-198, effectively works).
//   grid (i-blocks of 128 rows, skipped rows); W2: m x LD rows of L^-1; part[(k * nblk + iblk) * (D+1) + p]
// ---------------------------------------------------------------------------------------------
constexpr int LOWRANK_THREADS = 128, LOWRANK_CHUNK = 8;
template <int KT>
__global__ __launch_bounds__(LOWRANK_THREADS) void grad_lowrank_kernel(BatchView v, int emu, const double* __restrict__ W2, int nblk,
                                                                        double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ double red[LOWRANK_THREADS];
  const int n = v.n, D = v.D, ld = v.LD;
  const int k = blockIdx.y, tid = threadIdx.x, i = blockIdx.x * LOWRANK_THREADS + tid;
  const double* X = v.X + (size_t)emu * v.XS;
  const double* P = v.P + (size_t)emu * v.PS;
  const double* w = W2 + (size_t)k * ld;
  const double sig2 = P[D];
  double* sxi = sm;                                  // [d][tid]
  for (int d = 0; d < D; ++d) sxi[d * LOWRANK_THREADS + tid] = (i < n) ? X[(size_t)i * D + d] : 0.0;
  const double wi = (i < n) ? w[i] : 0.0;
  for (int pc = 0; pc <= D; pc += LOWRANK_CHUNK) {
    double acc[LOWRANK_CHUNK];
#pragma unroll
    for (int c = 0; c < LOWRANK_CHUNK; ++c) acc[c] = 0.0;
    for (int j = 0; j < n; ++j) {
      const double* xj = X + (size_t)j * D;
      const double wj = w[j];
      double g, kv;                                  // sigma^2 dk/dr2 (product kernel: sigma^2 k), sigma^2 k
      if (KT < 2) {
        double r2 = 0.0;
        for (int d = 0; d < D; ++d) {
          const double df = sxi[d * LOWRANK_THREADS + tid] - xj[d];
          r2 = __builtin_fma(P[d] * df, df, r2);
        }
        g = sig2 * kern_dr2<KT>(r2, EXP_TAB_G);
        kv = sig2 * kern_val<KT>(r2, EXP_TAB_G);
      } else {
        double kk = 1.0, ssum = 0.0;
        for (int d = 0; d < D; ++d) {
          const double df = sxi[d * LOWRANK_THREADS + tid] - xj[d];
          const double r2 = P[d] * df * df;
          const double sd = sqrt(5.0 * r2);
          kk *= 1.0 + sd + (5.0 / 3.0) * r2;
          ssum += sd;
        }
        kv = sig2 * (kk * lean_exp_neg<false>(ssum, EXP_TAB_G));
        g = kv;
      }
#pragma unroll
      for (int c = 0; c < LOWRANK_CHUNK; ++c) {
        const int p = pc + c;
        if (p < D) {
          const double df = sxi[p * LOWRANK_THREADS + tid] - xj[p];
          const double t = (KT < 2) ? g * (P[p] * df * df) : g * mat52_dlog(P[p] * df * df) * (P[p] * df * df);
          acc[c] = __builtin_fma(t, wj, acc[c]);
        } else if (p == D) {
          acc[c] = __builtin_fma(kv, wj, acc[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < LOWRANK_CHUNK; ++c) {
      const int p = pc + c;
      if (p > D) break;
      red[tid] = wi * acc[c];
      __syncthreads();
      for (int h = LOWRANK_THREADS / 2; h > 0; h >>= 1) {
        if (tid < h) red[tid] += red[tid + h];
        __syncthreads();
      }
      if (tid == 0) part[((size_t)k * nblk + blockIdx.x) * (D + 1) + p] = red[0];
      __syncthreads();
    }
  }
}

// out[emu][p] += 0.5 * sum of the partials, p <= D (fixed summation order)
__global__ void grad_lowrank_finish_kernel(int D, int count, const double* __restrict__ part, double* __restrict__ out) {
  const int p = threadIdx.x;
  if (p > D) return;
  double s = 0.0;
  for (int e = 0; e < count; ++e) s += part[(size_t)e * (D + 1) + p];
  out[p] += 0.5 * s;
}

void launch_grad_lowrank(const BatchView& v, int emu, const double* W2, int m, double* part, double* out, hipStream_t s) {
  const int nblk = (v.n + LOWRANK_THREADS - 1) / LOWRANK_THREADS;
  const size_t sm = (size_t)v.D * LOWRANK_THREADS * sizeof(double);
#define CALL(K) hipLaunchKernelGGL((grad_lowrank_kernel<K>), dim3(nblk, m), dim3(LOWRANK_THREADS), sm, s, v, emu, W2, nblk, part)
  KT_DISPATCH(v.kernel_type, CALL);
#undef CALL
  hipLaunchKernelGGL(grad_lowrank_finish_kernel, dim3(1), dim3(128), 0, s, v.D, nblk * m, part, out + (size_t)emu * (v.D + 3));
}

void launch_grad(const BatchView& v, double* partial, double* out, hipStream_t s) {
  const int ntiles = grad_num_tiles(v.n);
  const size_t sm = (size_t)(128 * v.D + 8 * 256) * sizeof(double);
  prof_begin("grad_reduce", s);
#define CALL(K) hipLaunchKernelGGL((grad_kernel<K>), dim3(ntiles, v.nb), dim3(256), sm, s, v, ntiles, partial)
  KT_DISPATCH(v.kernel_type, CALL);
#undef CALL
  prof_end("grad_reduce", s, 0., (double)v.nb * 4.0 * v.n * (double)v.n);
  hipLaunchKernelGGL(grad_finish_kernel, dim3(v.D + 3, v.nb), dim3(256), 0, s, v, ntiles, partial, out);
}

}  // namespace mogp
