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
#include <vector>

namespace mogp {

// Thread index for the device functions shared between the stand-alone kernels and the one-launch Cholesky.  In the
// latter (MOGP_OPAQUE_TID) the index is laundered through an empty asm statement: otherwise the compiler hoists every
// per-thread address computation of every task type out of the task loop and spills them (57 VGPRs).
#ifdef __HIPCC__
__device__ __forceinline__ int mogp_tid() {
  int t = threadIdx.x;
#ifdef MOGP_OPAQUE_TID
  asm volatile("" : "+v"(t));
  __builtin_assume(t >= 0 && t < 256);
#endif
  return t;
}
#endif

constexpr int TILE = 128;       // MFMA macro tile / padding granule
// The covariance kernels stage two 64-row blocks of X (and, for predict_deriv, a 64 x 65 work tile and a 64 x D
// accumulator) in LDS: (192 D + 4160) doubles <= 160 KB  ->  D <= 85.
constexpr int MAX_D = 80;
constexpr int NBI = 64;         // pivoted-Cholesky panel width / trtri leaf
constexpr double PAD_BIG = 1e300;
constexpr int RMAX = 8;         // max right-hand-side rows (targets + up to 7 mean-function columns)

struct BatchView {
  int n, D, NP, PS;             // PS = parameter block stride (doubles)
  int LD;                       // row stride of A / Linv / Kinv / Ks / alpha (= NP; padding measured slower)
  size_t MS;                    // per-emulator matrix stride = NP * LD
  int kernel_type;
  const double* X;              // n*D, or per emulator (XS = n*D) when the training inputs are held in pivoted order
  size_t XS = 0;                // per-emulator stride of X in doubles (0 = one matrix shared by all emulators)
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
// zero: up to two int ranges the launch also clears (the factorisation's info words, the one-launch Cholesky's control words) -- each was a
// memset command of its own in front of the kernel that needs it, 9 - 10 us per command on the path of a small fit
struct ZeroRanges {
  unsigned* p[2] = {nullptr, nullptr};
  unsigned n[2] = {0, 0};
};
void launch_cov_build(const BatchView& v, hipStream_t s, const ZeroRanges& zero = ZeroRanges());
// full symmetric K (no nugget) for get_K: out (n,n) for one emulator
void launch_cov_full(const BatchView& v, int emu, double* out, hipStream_t s);
// stand-alone kernel objects (bindings.cu:340-361): device buffers x1 (n1, D), x2 (n2, D), P = [exp(theta_d) (D), sigma^2];
// mode 0 K (n1, n2), 1 dK/dtheta (D+1, n1, n2), 2 dK/dx1 (n2, n1, D); kt = device kernel 0 / 1 / 2
void launch_kernel_object(int kt, const double* x1, int n1, const double* x2, int n2, int D, const double* P, int mode, double* out, hipStream_t s);
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
size_t lpack128_doubles_per_emulator();   // scratch written by the diagonal-block kernel, read by the panel solve
// 128 x 128 diagonal block at c0 (one workgroup per emulator, chol128_dev.h) and the 128-wide panel below it (MFMA block
// substitution); info[emu] = 0, or c0 + 1 of the FIRST 128-wide diagonal block that met a pivot that is not a positive finite
// number (the column inside the block is not recorded: a bad pivot turns into NaN, which reaches the block's last
// reciprocal square root, tested once -- chol128_dev.h; the engine only tests for non-zero)
void launch_panel128(const BatchView& v, int c0, int* info, double* Lpack128, hipStream_t s);
// C[i,j] -= sum_{k in [k0,k1)} A[i,k] A[j,k] for the 64-wide column block [c0,c0+64), rows [c0, NP)
void launch_update_narrow(const BatchView& v, int c0, int k0, int k1, hipStream_t s);
// the two 64-wide halves of the 128-wide column block [c0, c0+128) in one launch of 64 x 64 tiles
void launch_update_narrow_pair(const BatchView& v, int c0, int k0, int k1, hipStream_t s);
// same for a 128-wide column block with 128 x 128 tiles
void launch_update_wide(const BatchView& v, int c0, int k0, int k1, hipStream_t s);
// trailing lower-triangular update, rows/cols [c0, NP), k in [k0,k1) (c0 multiple of 128)
void launch_update_trailing(const BatchView& v, int c0, int k0, int k1, hipStream_t s);
// --- the whole blocked Cholesky as ONE launch (kernels_mchol.hip): persistent workgroups work off a dependency-ordered task
// queue.  table: mchol_task_table(NP) on the device; ctrl: mchol_ctrl_ints(NP, B) unsigned ints (zeroed by the launcher; ctrl[0]
// != 0 afterwards = a wait timed out and the factors are unusable: factorise again with a multi-launch schedule); packs:
// mchol_pack_doubles(NP, B) doubles (one diagonal-block pack per emulator and block column); info as launch_panel128.
// The matrix of one emulator must be smaller than 4 GB (write-through stores go through a buffer descriptor).
std::vector<int> mchol_task_table(int NP, bool ahead = false);
size_t mchol_ctrl_ints(int NP, int B);
size_t mchol_pack_doubles(int NP, int B);
// ctrl_zeroed: the caller has cleared ctrl[0 .. ctrl_ints) on the stream already (launch_cov_build's ZeroRanges)
// tables: the in-order table, then the band-ahead one (mchol_task_table(NP, true)), ntasks entries each
void launch_mchol(const BatchView& v, unsigned* ctrl, size_t ctrl_ints, const int* tables, int ntasks, double* packs, int* info, int n_cu,
                  hipStream_t s, bool ctrl_zeroed = false);
constexpr int MCHOL_ABORTED = -3;     // status reported by launch_logdet for every emulator when ctrl[0] != 0

// res (indexed by emulator, RES_STRIDE doubles each): [0] = 2 sum_{i<n} log L_ii, [1] = info[emu] -- or BACKSOLVE_TIMEOUT when the
// factorisation succeeded but bs_status[emu] == bs_epoch (the one-launch back substitution gave up waiting) --,
// [2 + r*RMAX + s] = sum_{c<n} L[n+r,c] L[n+s,c]  (Gram matrix of the right-hand-side rows; [2] = y^T y)
constexpr int RES_STRIDE = 2 + RMAX * RMAX;
constexpr int BACKSOLVE_TIMEOUT = -2;
void launch_logdet(const BatchView& v, const int* info, double* res, hipStream_t s, const int* bs_status = nullptr, int bs_epoch = 0,
                   const unsigned* mc_abort = nullptr);
// alpha[c] = sum_r M[emu][c][r] Z[r], c < RA   (M: indexed by emulator, (RMAX+1) x RMAX row-major)
void launch_combine_rows(const BatchView& v, const double* M, hipStream_t s);
// alpha = L^-T y (y = row n of A)
void launch_backsolve(const BatchView& v, hipStream_t s);
// the same in one launch (R == 1): flags = B * ceil(n/128) ints indexed by emulator, all != epoch on entry; status[emu] = epoch
// when a wait of that emulator's chain timed out (alpha is then unusable: repeat with launch_backsolve)
// info / res / mc_abort: as launch_logdet's -- the chain's leftmost chunk then writes res (log-determinant, Gram entry, status word) itself;
// returns whether it does (single right-hand side, sentinel form): the caller launches launch_logdet otherwise.
bool launch_backsolve_chain(const BatchView& v, int* flags, int epoch, int* status, int n_cu, hipStream_t s, const int* info = nullptr,
                            double* res = nullptr, const unsigned* mc_abort = nullptr);
// Pivoted Cholesky with LAPACK dpstrf semantics (nugget="pivot", linalg/cholesky.py:284-327), see kernels_pivot.hip.
// A (K without nugget, and the right-hand-side rows) is factored in place with symmetric row/column interchanges:
//   begin -> { panel(k0, 64 columns) -> rank-64 update of everything to its right } ... -> tail -> end
// (after every panel the host drops the emulators whose factorisation stopped)
// perm[emu*n + pos] = original index of the point now at position pos; rank[emu] = -1 while running, else the number of
// pivots accepted; info[emu] = 1 when no pivot is positive.  work: B * pstrf_work_doubles(NP) doubles, indexed by emulator.
__host__ __device__ inline size_t pstrf_work_doubles(int NP) { return 2 * (size_t)NP + 8; }
void launch_pstrf_begin(const BatchView& v, int* perm, int* rank, int* info, double* work, hipStream_t s);
void launch_pstrf_panel(const BatchView& v, int k0, int jb, int* perm, int* rank, double* work, hipStream_t s);   // jb <= 64
// emulators of the launch with 0 < rank < n: restore the skipped block from the kernel function on X0 (training order) or
// from the input matrix A0 (n x n), replacement diagonal, right-hand-side rows through the skipped block
void launch_pstrf_tail(const BatchView& v, const int* perm, const int* rank, const double* X0, const double* A0, hipStream_t s);
void launch_pstrf_end(const BatchView& v, hipStream_t s);
// Xp[emu] (n, D) <- X[perm[emu]] for the slots of the launch
void launch_permute_rows(const BatchView& v, const double* X, const int* perm, double* Xp, hipStream_t s);

// --- L^-1, K^-1 ----------------------------------------------------------------------------
void launch_trtri(const BatchView& v, hipStream_t s);        // A(L) -> Linv   (uses Kinv as scratch)
void launch_kinv(const BatchView& v, int n_cu, hipStream_t s);   // Linv -> Kinv (lower tiles); n_cu: compute units of the engine's device
void launch_alpha_from_linv(const BatchView& v, hipStream_t s);   // alpha = Linv^T y

// --- gradient ------------------------------------------------------------------------------
// partial: nb * ntiles * (D+3) doubles scratch; out: per emulator (D+3): [g_0..g_{D-1}, g_cov, tr(Kinv), alpha.alpha]
int grad_num_tiles(int n);
void launch_grad(const BatchView& v, double* partial, double* out, hipStream_t s);
// rank-deficient pivoted factor: out[emu][p] += 0.5 sum_k w_k^T dK_p w_k over the m rows W2 (m x LD) of L^-1 that K^-1 was
// formed without; part: m * ceil(n/128) * (D+1) doubles scratch
void launch_grad_lowrank(const BatchView& v, int emu, const double* W2, int m, double* part, double* out, hipStream_t s);

// --- predict -------------------------------------------------------------------------------
// Ks: nb * MP * NP (MP = roundup(m,128)) cross-covariance sigma^2 k(x*_m, x_j); mean (nb, m) written;
// if Ks == null only the mean is computed.  With R > 1 the kernel also returns Z_c^T k* (rows 1..R-1 of `mean`,
// row stride mean_ld per emulator block of R rows).
void launch_cross_cov_mean(const BatchView& v, const double* Xs, int m, int MP, double* Ks, double* mean, int mean_ld, hipStream_t s);
// var[z][m] = sigma^2 - sum_i (Linv Ks^T)[i][m]^2 ; partial: nb * (NP/128) * MP scratch
void launch_predict_var(const BatchView& v, const double* Ks, int m, int MP, double* partial, double* var, int var_ld, int n_cu, hipStream_t s);
// deriv[z][m][d]
void launch_predict_deriv(const BatchView& v, const double* Xs, int m, double* deriv, long deriv_stride, hipStream_t s);

// mean-function terms of a batched prediction on device buffers (kernels_chol.hip predict_mean_finish_kernel): basis (nbasis, m),
// coef (nb, nbasis), R > 1: dots (nb, R, m) and LA (nb, q, q); mean / var (nb rows of stride ld; either may be null); derivative terms
// dbasis (nterm, m), ddims / dpowers (nterm) into deriv (nb, m, D) or null
// basis (1 + nterm, m) / dbasis (nterm, m) of a polynomial mean at device-resident points Xs (m, D); dims / powers: device ints
void launch_mean_basis(const double* Xs, int m, int D, int nterm, const int* dims, const int* powers, double* basis, double* dbasis, hipStream_t s);
void launch_predict_mean_finish(int nb, int m, int D, int R, int nbasis, const double* basis, const double* coef, const double* dots,
                                const double* LA, double* mean, double* var, long ld, int nterm, const double* dbasis, const int* ddims,
                                const int* dpowers, double* deriv, hipStream_t s);

// --- utilities -------------------------------------------------------------------------------
// out (n,n) <- tile of src (NP,NP): mode 0 copy, mode 1 transpose, mode 2 symmetrise from lower
void launch_extract(const double* src, int NP, int n, double* out, int mode, hipStream_t s);

// --- profiling hooks (bench only) -------------------------------------------------------------
bool prof_is_on();
void prof_begin(const char* tag, hipStream_t s);
void prof_end(const char* tag, hipStream_t s, double flops, double bytes);

}  // namespace mogp
