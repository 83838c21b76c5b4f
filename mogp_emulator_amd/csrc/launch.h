// Host-side launchers for the gfx950 kernels.  All launches are asynchronous on `stream`.
//
// Data layout in HBM:
//   X      (n, D) row-major, shared by every emulator of an engine
//   P      per emulator parameter block of PS doubles: [e_0..e_{D-1} = exp(theta_d), sigma^2, nugget]
//   T      per emulator residual targets (n)
//   A      per emulator NP x NP "factor" matrix, row-major with row stride LD (= NP), NP = roundup(n+1, 128).
//          rows/cols < n : K + nugget I -> overwritten in place by L (lower triangle)
//          row n         : the targets t -> overwritten by y^T = (L^-1 t)^T   (free forward solve)
//          rows > n      : identity padding
//   Linv   per emulator NP x NP row-major, lower triangular L^-1
//   Kinv   per emulator NP x NP row-major, K^-1 (lower tiles valid); doubles as scratch for trtri
//   batch slot z -> emulator index: idx ? idx[z] : z
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mogp {

constexpr int TILE = 128;       // MFMA macro tile / padding granule
// The covariance kernels stage two 64-row blocks of X (and, for predict_deriv, a 64 x 65 work tile and a 64 x D
// accumulator) in LDS: (192 D + 4160) doubles <= 160 KB  ->  D <= 85.
constexpr int MAX_D = 80;
constexpr int NBI = 64;         // inner panel width (potf2 / trsm / trtri leaf)
constexpr double PAD_BIG = 1e300;
constexpr int RMAX = 8;         // max right-hand-side rows (targets + up to 7 mean-function columns)

struct BatchView {
  int n, D, NP, PS;             // PS = parameter block stride (doubles)
  int LD;                       // row stride of A / Linv / Kinv / Ks / alpha (= NP; padding measured slower)
  size_t MS;                    // per-emulator matrix stride = NP * LD
  int kernel_type;
  const double* X;              // n*D
  const double* P;              // B*PS
  const double* T;              // B*n
  double* A;                    // B*NP*NP
  double* Linv;                 // B*NP*NP (may be null)
  double* Kinv;                 // B*NP*NP (may be null)
  double* alpha;                // B*RA*LD.  R = 1: the single row K^-1 t.  R > 1 (analytic mean): row 0 = K^-1 (t - H beta_hat) (predictions),
                                // rows 1..q = rank-correction rows g_c, row R = K^-1 (t - H (b + beta')) (gradient; equals row 0 with weak mean priors)
  double* Z;                    // B*R*LD raw solves K^-1 [t, h_1 .. h_q]   (same buffer as alpha when R = 1)
  const double* H;              // q*n design-matrix columns of the analytic mean (shared by all emulators), or null
  int R;                        // 1 + q right-hand sides carried through the factorisation as rows n .. n+q of A
  int RA;                       // rows per emulator in alpha: 1 (R = 1) or R + 1
  const int* idx;               // device: nb entries or null
  int nb;                       // number of batch slots in this launch
};

// --- covariance build ---------------------------------------------------------------------
void launch_cov_build(const BatchView& v, hipStream_t s);
// full symmetric K (no nugget) for get_K: out (n,n) for one emulator
void launch_cov_full(const BatchView& v, int emu, double* out, hipStream_t s);
// leave-one-out predictive variance 1/[K^-1]_ii of every training input (needs Linv): out[slot*out_ld + i]
void launch_loo_variance(const BatchView& v, double* out, int out_ld, hipStream_t s);
// history-matching score of m query points from device-resident means / variances (nb, ld); prm (nb, 3) =
// [observation, obsvar + discrepancy + nugget, mean offset]; out[j] = (rank+1)-th largest implausibility
constexpr int IMPLAUS_MAX_RANK = 15;
void launch_implausibility(int nb, const double* mean, const double* var, int ld, int m, const double* prm, int rank, double* out,
                           hipStream_t s);
// out (nb, m, m) = sigma^2 k(Xs, Xs) per slot (no nugget)
void launch_cov_self_batch(const BatchView& v, const double* Xs, int m, double* out, hipStream_t s);
// full predictive covariance: cov (nb, m, m) holds K** on entry, K** - Ks K^-1 Ks^T on return; V: nb*NP*MP scratch
void launch_predict_fullcov(const BatchView& v, const double* Ks, int m, int MP, double* V, double* cov, hipStream_t s);

// --- blocked Cholesky ---------------------------------------------------------------------
// potf2 of the 64x64 diagonal block at c0; info[emu] = first failing (1-based) column or 0
void launch_potf2(const BatchView& v, int c0, int* info, double* Lpack, hipStream_t s);
size_t lpack_doubles_per_emulator();   // scratch written by potf2, read by trsm
size_t lpack128_doubles_per_emulator();
// 128 x 128 diagonal block (potf2 + trsm + update + potf2 in one workgroup per emulator) and the panel below it
void launch_panel128(const BatchView& v, int c0, int* info, double* Lpack128, hipStream_t s);
// rows [r0, NP) of column block [c0, c0+64): X L_kk^T = A
void launch_trsm(const BatchView& v, int c0, int r0, const double* Lpack, hipStream_t s);
// C[i,j] -= sum_{k in [k0,k1)} A[i,k] A[j,k] for the 64-wide column block [c0,c0+64), rows [c0, NP)
void launch_update_narrow(const BatchView& v, int c0, int k0, int k1, hipStream_t s);
// same, and the diagonal-tile workgroup then factors the 64x64 block at (c0,c0) (fused potf2)
void launch_update_narrow_potf2(const BatchView& v, int c0, int k0, int k1, int* info, double* Lpack, hipStream_t s);
// same for a 128-wide column block (MFMA 128x128 tiles)
// Role-fused step launch: up to two jobs over disjoint emulator groups (see kernels_gemm.hip)
enum { ROLE_UPDATE = 0, ROLE_POTF2 = 1, ROLE_TRSM = 2 };
struct FusedJob {
  int role;
  int wg_begin, wg_count;   // workgroup range of the job inside the launch (wg_begin is a multiple of 8)
  int idx_off, nb;          // emulator group: idx[idx_off .. idx_off + nb)
  int per_emu;              // workgroups per emulator (tiles of this slice / row blocks / 1)
  int c0, k0, k1, nt, tile0, r0;
};
struct FusedArgs {
  FusedJob job[2];
  int njobs;
};
void launch_fused_step(const BatchView& v, const FusedArgs& fa, int total_wgs, int* info, double* Lpack, hipStream_t s);
void launch_update_narrow_pair(const BatchView& v, int c0, int k0, int k1, hipStream_t s);
void launch_update_wide(const BatchView& v, int c0, int k0, int k1, hipStream_t s);
// trailing lower-triangular update, rows/cols [c0, NP), k in [k0,k1) (c0 multiple of 128)
void launch_update_trailing(const BatchView& v, int c0, int k0, int k1, hipStream_t s);
// logdet[emu] = 2 sum_{i<n} log L_ii ; gram[emu][r*R+s] = sum_{c<n} L[n+r,c] L[n+s,c]  (gram[0] = y^T y)
void launch_logdet(const BatchView& v, double* logdet, double* gram, hipStream_t s);
// alpha[c] = sum_r M[emu][c][r] Z[r], c < RA   (M: indexed by emulator, (RMAX+1) x RMAX row-major)
void launch_combine_rows(const BatchView& v, const double* M, hipStream_t s);
// alpha = L^-T y (y = row n of A)
void launch_backsolve(const BatchView& v, hipStream_t s);

// --- L^-1, K^-1 ----------------------------------------------------------------------------
void launch_trtri(const BatchView& v, hipStream_t s);        // A(L) -> Linv   (uses Kinv as scratch)
void launch_kinv(const BatchView& v, hipStream_t s);         // Linv -> Kinv (lower tiles)
void launch_alpha_from_linv(const BatchView& v, hipStream_t s);   // alpha = Linv^T y

// --- gradient ------------------------------------------------------------------------------
// partial: nb * ntiles * (D+3) doubles scratch; out: per emulator (D+3): [g_0..g_{D-1}, g_cov, tr(Kinv), alpha.alpha]
int grad_num_tiles(int n);
void launch_grad(const BatchView& v, double* partial, double* out, hipStream_t s);

// --- predict -------------------------------------------------------------------------------
// Ks: nb * MP * NP (MP = roundup(m,128)) cross-covariance sigma^2 k(x*_m, x_j); mean (nb, m) written;
// if Ks == null only the mean is computed.  With R > 1 the kernel also returns Z_c^T k* (rows 1..R-1 of `mean`,
// row stride mean_ld per emulator block of R rows).
void launch_cross_cov_mean(const BatchView& v, const double* Xs, int m, int MP, double* Ks, double* mean, int mean_ld, hipStream_t s);
// var[z][m] = sigma^2 - sum_i (Linv Ks^T)[i][m]^2 ; partial: nb * (NP/128) * MP scratch
void launch_predict_var(const BatchView& v, const double* Ks, int m, int MP, double* partial, double* var, int var_ld, hipStream_t s);
// deriv[z][m][d]
void launch_predict_deriv(const BatchView& v, const double* Xs, int m, double* deriv, long deriv_stride, hipStream_t s);

// --- utilities -------------------------------------------------------------------------------
// out (n,n) <- tile of src (NP,NP): mode 0 copy, mode 1 transpose, mode 2 symmetrise from lower
void launch_extract(const double* src, int NP, int n, double* out, int mode, hipStream_t s);

// --- profiling hooks (bench only) -------------------------------------------------------------
bool prof_is_on();
void prof_begin(const char* tag, hipStream_t s);
void prof_end(const char* tag, hipStream_t s, double flops, double bytes);

}  // namespace mogp
