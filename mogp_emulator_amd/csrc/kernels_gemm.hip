// fp64 MFMA tile kernels for gfx950 (v_mfma_f64_16x16x4_f64).
//
// One device main loop, gemm_mainloop<WT, A_KMAJ, B_KMAJ>, computes for a workgroup tile
//     acc(i, j) = sum_k opA(i, k) * opB(j, k)
// where each operand is either "K-major" (stored [row][k], k contiguous) or "M-major"
// (stored [k][row], row contiguous).  Both natural global layouts are kept unchanged in LDS:
//   K-major LDS tile  [BM][16+2]   : fragment read (row = lane&15, k = lane>>4) hits 32 distinct
//                                    8-byte bank pairs per 32-lane half (row stride 18 doubles)
//   M-major LDS tile  [16][BM+16]  : same, because the k stride is == 16 (mod 32) doubles
// so there is no transposing copy anywhere and every ds_read_b64 is conflict free.
// Workgroup = 256 threads = 4 waves in a 2x2 grid; each wave owns WT x WT MFMA tiles of 16x16
// (WT=4: 128x128 block tile, 128 accumulator VGPRs; WT=2: 64x64 block tile).
// The K loop is double buffered in LDS with a register-staged global prefetch (one barrier per
// 16-deep step; 64 MFMAs per wave per step at WT=4).
//
// fp64 C/D fragment map (verified on hardware by tools/mfma_probe.hip):
//   row = (lane>>4) + 4*reg, col = lane&15;  A: A[lane&15][lane>>4];  B: B[lane>>4][lane&15].
//
// Reference operations these kernels replace (as a group): cusolverDnDpotrf's trailing updates
// (densegp_gpu.hpp:451-474), the explicit inverse via potrs (densegp_gpu.hpp:576-582) and the
// predictive-variance gemm + batched dot (densegp_gpu.hpp:374-396).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "launch.h"
#include "trsm_dev.h"
#include "gemm_dev.h"

namespace mogp {


template <int WT>
struct Cfg {
  static constexpr int BM = 32 * WT;            // block tile edge
  static constexpr int WM = 16 * WT;            // wave tile edge
  static constexpr int LDM = BM + 16;
  static constexpr int CH = BM * 8 / 256;       // 16-byte chunks per thread per operand tile
  static constexpr int OPSZ = (BM * LDK > BK * LDM) ? BM * LDK : BK * LDM;
  static constexpr int SMEM_DOUBLES = 4 * OPSZ; // 2 buffers x (A, B)
};

// Global -> register staging of one operand tile.  A thread's chunks q = 0 .. CH-1 lie a whole number of rows apart, so it needs ONE
// 32-bit byte offset (g2r_off) against wave-uniform bases g + q * ROWS * ld: scalar base + vector offset addressing, no 64-bit address
// arithmetic per load and fewer address registers than per-thread pointers (predictive variance: 65.4 -> 66.0 TFLOP/s with the same change).
template <int WT, bool KMAJ>
__device__ __forceinline__ unsigned g2r_off(int ld) {
  const int t = threadIdx.x;
  const int off = KMAJ ? (t >> 3) * ld + (t & 7) * 2 : (t / (Cfg<WT>::BM / 2)) * ld + (t % (Cfg<WT>::BM / 2)) * 2;
  return (unsigned)(off * (int)sizeof(double));
}
template <int WT, bool KMAJ>
__device__ __forceinline__ void g2r(const double* __restrict__ g, int ld, unsigned off, v2d (&r)[Cfg<WT>::CH]) {
  constexpr int ROWS = KMAJ ? 32 : 256 / (Cfg<WT>::BM / 2);     // rows between a thread's consecutive chunks
#pragma unroll
  for (int q = 0; q < Cfg<WT>::CH; ++q) r[q] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(g + (size_t)q * ROWS * ld) + off);
}

template <int WT, bool KMAJ>
__device__ __forceinline__ void r2s(double* s, const v2d (&r)[Cfg<WT>::CH]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < Cfg<WT>::CH; ++q) {
    const int c = t + 256 * q;
    if (KMAJ) {
      const int row = c >> 3, kc = (c & 7) * 2;
      *reinterpret_cast<v2d*>(s + row * LDK + kc) = r[q];
    } else {
      const int krow = c / (Cfg<WT>::BM / 2), mc = (c % (Cfg<WT>::BM / 2)) * 2;
      *reinterpret_cast<v2d*>(s + krow * Cfg<WT>::LDM + mc) = r[q];
    }
  }
}

template <int WT, bool KMAJ>
__device__ __forceinline__ double frag(const double* s, int row, int k) {
  return KMAJ ? s[row * LDK + k] : s[k * Cfg<WT>::LDM + row];
}

// Ag: K-major -> &A[i0*lda + k0] ; M-major -> &A[k0*lda + i0].  nk = number of 16-deep steps.
template <int WT, bool AK, bool BKM>
__device__ __forceinline__ void gemm_mainloop(const double* __restrict__ Ag, int lda, const double* __restrict__ Bg,
                                              int ldb, int nk, v4d (&acc)[WT][WT], double* smem) {
  using C = Cfg<WT>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) acc[i][j] = (v4d){0., 0., 0., 0.};
  if (nk <= 0) return;

  const size_t stepA = AK ? (size_t)BK : (size_t)BK * lda;
  const size_t stepB = BKM ? (size_t)BK : (size_t)BK * ldb;
  v2d ra[C::CH], rb[C::CH];
  const unsigned offA = g2r_off<WT, AK>(lda), offB = g2r_off<WT, BKM>(ldb);
  g2r<WT, AK>(Ag, lda, offA, ra);
  g2r<WT, BKM>(Bg, ldb, offB, rb);
  r2s<WT, AK>(smem, ra);
  r2s<WT, BKM>(smem + C::OPSZ, rb);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const double* sA = smem + (kt & 1) * 2 * C::OPSZ;
    const double* sB = sA + C::OPSZ;
    const bool more = (kt + 1 < nk);
    if (more) {
      Ag += stepA;
      Bg += stepB;
      g2r<WT, AK>(Ag, lda, offA, ra);
      g2r<WT, BKM>(Bg, ldb, offB, rb);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a[WT], b[WT];
      const int k = kk * 4 + fk;
#pragma unroll
      for (int i = 0; i < WT; ++i) a[i] = frag<WT, AK>(sA, wr * C::WM + i * 16 + fr, k);
#pragma unroll
      for (int j = 0; j < WT; ++j) b[j] = frag<WT, BKM>(sB, wc * C::WM + j * 16 + fr, k);
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      double* dA = smem + ((kt + 1) & 1) * 2 * C::OPSZ;
      r2s<WT, AK>(dA, ra);
      r2s<WT, BKM>(dA + C::OPSZ, rb);
    }
    __syncthreads();
  }
}

// iterate the accumulator fragment: f(row_in_tile, col_in_tile, value&)
template <int WT, typename F>
__device__ __forceinline__ void for_each_acc(v4d (&acc)[WT][WT], F f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        f(wr * Cfg<WT>::WM + i * 16 + (lane >> 4) + 4 * r, wc * Cfg<WT>::WM + j * 16 + (lane & 15), acc[i][j][r]);
}

__device__ __forceinline__ int slot_to_emu(const int* idx, int z) { return idx ? idx[z] : z; }

// Decode a 1-D grid into (batch slot z, tile id).  When the slot count is a multiple of 8 each
// XCD (block id % 8) works through whole emulators one after another, so an emulator's panels
// and trailing matrix stay inside one 4 MiB L2 instead of being spread over all eight.
__device__ __forceinline__ void decode_block(int nb, int ntiles, int& z, int& tile, int L = (int)blockIdx.x) {
  if ((nb & 7) == 0) {
    const int xcd = L & 7, w = L >> 3;
    z = (w / ntiles) * 8 + xcd;
    tile = w % ntiles;
  } else {
    z = L / ntiles;
    tile = L % ntiles;
  }
}

// ---------------------------------------------------------------------------------------------
// Narrow symmetric update on A:  C[i,j] -= sum_{k in [k0,k1)} A[i,k] A[j,k] for a tile column [c0, c0+BM) (or the pair of 64-wide tile
// columns of a 128-wide block column), rows [c0, NP).  (The lower-triangular tile set of the trailing update is update_tri8_kernel; the
// 2 x 2-wave form of it that lived here behind MOGP_TRI_WAVES=4 went in round 6.)
// ---------------------------------------------------------------------------------------------
template <int WT>
__global__ __launch_bounds__(256, 2) void update_kernel(BatchView v, int c0, int k0, int k1, int nt, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using C = Cfg<WT>;
  int z, tile;
  decode_block(v.nb, ntiles, z, tile);
  if (z >= v.nb) return;
  int ti, tj;
  if (ntiles == nt) {
    ti = tile;
    tj = 0;
  } else {
    // pair launch (ntiles = 2 nt - 1): tile 0 = (0, 0), then the two tiles of a row tile next to each other --
    // (ti, 0) and (ti, 1) read the same 64 x K row panel, and as neighbours in dispatch order (same XCD) the second one
    // finds it in L2 instead of fetching it again (the column-by-column order had nt - 1 other tiles in between)
    ti = (tile + 1) >> 1;
    tj = (tile > 0 && (tile & 1) == 0) ? 1 : 0;
  }
  const int emu = slot_to_emu(v.idx, z);
  double* A = v.A + (size_t)emu * v.MS;
  const int ld = v.LD;
  const int i0 = c0 + ti * C::BM, j0 = c0 + tj * C::BM;
  v4d acc[WT][WT];
  gemm_mainloop<WT, true, true>(A + (size_t)i0 * ld + k0, ld, A + (size_t)j0 * ld + k0, ld, (k1 - k0) / BK, acc, smem);
  for_each_acc<WT>(acc, [&](int r, int c, double x) {
    double* p = A + (size_t)(i0 + r) * ld + (j0 + c);
    *p -= x;
  });
}

// 128-wide panel solve below a factored 128 x 128 diagonal block (trsm_dev.h): one wave per 16-row slab, four per workgroup
__global__ __launch_bounds__(256, 3) void trsm128_lds_kernel(BatchView v, int c0, int r0, const double* __restrict__ Lpack128) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int emu = slot_to_emu(v.idx, blockIdx.y);
  trsm128_lds_dev(v, c0, r0, Lpack128 + (size_t)emu * PACK128_STRIDE, emu, blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// trtri merge, level h:  node q covers [base, base+2h), base = q*2h
//   STEP 0:  T     = L21 * Linv11          (T kept in scratch at the position of block 21)
//   STEP 1:  Linv21 = -Linv22 * T
// Round 4: both products skip the structural zeros of their triangular operand at the 16 x 16 sub-tile level (merge_mainloop), as the
// predictive variance and K^-1 do, and the upper levels can use 128 x 128 tiles (launch_trtri_merges).
// ---------------------------------------------------------------------------------------------
// A K-major, B M-major, 2 x 2 waves, sub-tiles dealt round-robin (row sub-tile 2 i + wr, column sub-tile 2 j + wc).
// MODE 0 (T = L21 Linv11, k from j0): the first BM values of k are the DIAGONAL block of Linv11 [k][j]: column sub-tile b has non-zeros
//         from its step kd = b on -- the set of active column sub-tiles [0, E) GROWS.
// MODE 1 (Linv21 = -Linv22 T, k up to i0 + BM): the LAST nd steps are the diagonal block of Linv22 [i][k]: row sub-tile a has non-zeros up
//         to step a of that block -- the set of active row sub-tiles [S, WT) SHRINKS (nk_full = steps in front of the block).
// The k loop is cut into consecutive loops, one per set; skipped products are exact zeros (entries unchanged bit for bit).
template <int WT, int MODE>
__device__ __forceinline__ void merge_mainloop(const double* __restrict__ Ag, int lda, const double* __restrict__ Bg, int ldb, int nk, int nk_full,
                                               v4d (&acc)[WT][WT], double* smem, int wr, int wc) {
  using C = Cfg<WT>;
  const int lane = threadIdx.x & 63;
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) acc[i][j] = (v4d){0., 0., 0., 0.};
  if (nk <= 0) return;
  const size_t stepB = (size_t)BK * ldb;
  v2d ra[C::CH], rb[C::CH];
  const unsigned offA = g2r_off<WT, true>(lda), offB = g2r_off<WT, false>(ldb);
  g2r<WT, true>(Ag, lda, offA, ra);
  g2r<WT, false>(Bg, ldb, offB, rb);
  r2s<WT, true>(smem, ra);
  r2s<WT, false>(smem + C::OPSZ, rb);
  __syncthreads();
  // one k-step with the row sub-tiles [S, WT) and the column sub-tiles [0, E)
  auto step = [&](int kt, auto S_, auto E_) {
    constexpr int S = decltype(S_)::value, E = decltype(E_)::value;
    const double* sA = smem + (kt & 1) * 2 * C::OPSZ;
    const double* sB = sA + C::OPSZ;
    const bool more = (kt + 1 < nk);
    if (more) {
      Ag += BK;
      Bg += stepB;
      g2r<WT, true>(Ag, lda, offA, ra);
      g2r<WT, false>(Bg, ldb, offB, rb);
    }
    if (S < WT && E > 0) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double a[WT], b[WT];
        const int k = kk * 4 + fk;
#pragma unroll
        for (int i = S; i < WT; ++i) a[i] = sA[((2 * i + wr) * 16 + fr) * LDK + k];
#pragma unroll
        for (int j = 0; j < E; ++j) b[j] = sB[k * C::LDM + (2 * j + wc) * 16 + fr];
#pragma unroll
        for (int i = S; i < WT; ++i)
#pragma unroll
          for (int j = 0; j < E; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      double* dA = smem + ((kt + 1) & 1) * 2 * C::OPSZ;
      r2s<WT, true>(dA, ra);
      r2s<WT, false>(dA + C::OPSZ, rb);
    }
    __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>;
  using IW = std::integral_constant<int, WT>;
  int kt = 0;
  static_assert(WT == 2 || WT == 4, "phase lists are written for two or four sub-tiles per wave");
  if (MODE == 0) {
    // column sub-tile 2 j + wc is active from step 2 j + wc on
    for (const int end = min(nk, wc); kt < end; ++kt) step(kt, I0(), I0());
    for (const int end = min(nk, 2 + wc); kt < end; ++kt) step(kt, I0(), std::integral_constant<int, 1>());
    if (WT == 4) {
      for (const int end = min(nk, 4 + wc); kt < end; ++kt) step(kt, I0(), std::integral_constant<int, 2>());
      for (const int end = min(nk, 6 + wc); kt < end; ++kt) step(kt, I0(), std::integral_constant<int, 3>());
    }
    for (; kt < nk; ++kt) step(kt, I0(), IW());
  } else {
    // row sub-tile 2 i + wr is active up to step nk_full + 2 i + wr
    for (const int end = min(nk, nk_full + wr + 1); kt < end; ++kt) step(kt, I0(), IW());
    for (const int end = min(nk, nk_full + 2 + wr + 1); kt < end; ++kt) step(kt, std::integral_constant<int, 1>(), IW());
    if (WT == 4) {
      for (const int end = min(nk, nk_full + 4 + wr + 1); kt < end; ++kt) step(kt, std::integral_constant<int, 2>(), IW());
      for (const int end = min(nk, nk_full + 6 + wr + 1); kt < end; ++kt) step(kt, std::integral_constant<int, 3>(), IW());
    }
    for (; kt < nk; ++kt) step(kt, IW(), IW());
  }
}

template <int WT, int STEP>
__global__ __launch_bounds__(256, 2) void trtri_merge_kernel(BatchView v, int h, int tiles_per_dim, int nodes) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using C = Cfg<WT>;
  // 1-D grid decoded so that each XCD walks through whole emulators (decode_block): the tiles of a node share
  // their operand panels through one L2 instead of being dealt round-robin over all eight
  const int tpn = tiles_per_dim * tiles_per_dim;
  int z, t;
  decode_block(v.nb, nodes * tpn, z, t);
  if (z >= v.nb) return;
  const int node = t / tpn, bx = t % tpn;
  const int emu = slot_to_emu(v.idx, z);
  const int ld = v.LD;
  const int base = node * 2 * h;
  if (base + h >= v.NP) return;
  const int m2 = min(h, v.NP - base - h);
  // longest-K tiles are dispatched first (STEP 0: K = h - j0 -> small tj first; STEP 1: K = i0 + BM ->
  // large ti first) so the tail of the launch is made of short tiles
  int ti, tj;
  if (STEP == 0) {
    tj = bx / tiles_per_dim;
    ti = bx % tiles_per_dim;
  } else {
    ti = tiles_per_dim - 1 - (bx / tiles_per_dim);
    tj = bx % tiles_per_dim;
  }
  const int i0 = ti * C::BM, j0 = tj * C::BM;          // node-local
  if (i0 >= m2) return;
  const double* L = v.A + (size_t)emu * v.MS;
  double* Li = v.Linv + (size_t)emu * v.MS;
  double* S = v.Kinv + (size_t)emu * v.MS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  v4d acc[WT][WT];
  // entry (row sub-tile 2 i + wr, column sub-tile 2 j + wc) of the tile
  auto store = [&](double* dst, double sign) {
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dst[(size_t)((2 * i + wr) * 16 + (lane >> 4) + 4 * r) * ld + (2 * j + wc) * 16 + (lane & 15)] = sign * acc[i][j][r];
  };
  if (STEP == 0) {
    // T[i,j] = sum_{k >= j0} L21[i,k] Linv11[k,j]
    const double* Ag = L + (size_t)(base + h + i0) * ld + base + j0;       // K-major, k from j0
    const double* Bg = Li + (size_t)(base + j0) * ld + base + j0;          // M-major: [k][j], k from j0
    merge_mainloop<WT, 0>(Ag, ld, Bg, ld, (h - j0) / BK, 0, acc, smem, wr, wc);
    store(S + (size_t)(base + h + i0) * ld + base + j0, 1.0);
  } else {
    // Linv21[i,j] = - sum_{k < i0+BM} Linv22[i,k] T[k,j]
    const double* Ag = Li + (size_t)(base + h + i0) * ld + base + h;       // K-major, k from 0
    const double* Bg = S + (size_t)(base + h) * ld + base + j0;            // M-major: T[k][j]
    const int kend = min(i0 + C::BM, m2);
    merge_mainloop<WT, 1>(Ag, ld, Bg, ld, kend / BK, i0 / BK, acc, smem, wr, wc);
    store(Li + (size_t)(base + h + i0) * ld + base + j0, -1.0);
  }
}

// ---------------------------------------------------------------------------------------------
// Kinv = Linv^T Linv (lower tiles):  Kinv[i,j] = sum_{k >= i0} Linv[k,i] Linv[k,j]
// ---------------------------------------------------------------------------------------------
// Structural zeros (round 4).  The k range of tile (ti, tj) starts at i0 = 128 ti, i.e. with the DIAGONAL block of L^-1's row panel:
// in its k-step kd = 0 .. 7 the A operand L^-1[k][i0 + row] is zero for the 16-row sub-tiles a > kd (28 of the 64 (sub-tile, step)
// pairs), and of a diagonal tile only the sub-tile pairs a >= b are ever read (gradient reduction, get_invQ: entries j <= i).  The
// dense kernel issued these products all the same: 0.844 MFMA-busy at 0.69 of the peak in algorithmic flops.  As in the
// predictive variance (mainloop_w<.., TRIA>) the sub-tiles are dealt to the wave rows / columns ROUND-ROBIN (a = 2 i + wr, b = 2 j + wc),
// so that every wave owns early and late sub-tiles, and the k loop is cut into CONSECUTIVE loops, one per set of active row sub-tiles
// [0, E) -- each the dense straight-line step restricted to that set; waves meet at the step barriers whatever their set.
// TRI: 0 every column sub-tile; 1 pairs j <= i; 2 pairs j < i (diagonal tiles: b <= a for the wave's residues).
// All skipped products are exact zeros or never read: the sums of the products that remain are unchanged (bit-identical entries).
template <int WT, int TRI>
__device__ __forceinline__ void kinv_mainloop(const double* __restrict__ Ag, const double* __restrict__ Bg, int ld, int nk, v4d (&acc)[WT][WT],
                                              double* smem, int wr, int wc) {
  using C = Cfg<WT>;
  const int lane = threadIdx.x & 63;
  const int fr = lane & 15, fk = lane >> 4;
  const size_t stepA = (size_t)BK * ld;
  v2d ra[C::CH], rb[C::CH];
  const unsigned offA = g2r_off<WT, false>(ld);
  g2r<WT, false>(Ag, ld, offA, ra);
  g2r<WT, false>(Bg, ld, offA, rb);
  r2s<WT, false>(smem, ra);
  r2s<WT, false>(smem + C::OPSZ, rb);
  __syncthreads();
  auto step = [&](int kt, auto E_) {
    constexpr int E = decltype(E_)::value;
    const double* sA = smem + (kt & 1) * 2 * C::OPSZ;
    const double* sB = sA + C::OPSZ;
    const bool more = (kt + 1 < nk);
    if (more) {
      Ag += stepA;
      Bg += stepA;
      g2r<WT, false>(Ag, ld, offA, ra);
      g2r<WT, false>(Bg, ld, offA, rb);
    }
    constexpr int JN = (TRI == 0) ? WT : ((TRI == 1) ? E : E - 1);      // column sub-tiles any active row sub-tile pairs with
    if (E > 0 && JN > 0) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double a[WT], b[WT];
        const int k = kk * 4 + fk;
#pragma unroll
        for (int i = 0; i < E; ++i) a[i] = sA[k * C::LDM + (2 * i + wr) * 16 + fr];
#pragma unroll
        for (int j = 0; j < JN; ++j) b[j] = sB[k * C::LDM + (2 * j + wc) * 16 + fr];
#pragma unroll
        for (int i = 0; i < E; ++i)
#pragma unroll
          for (int j = 0; j < JN; ++j)
            if (TRI == 0 || (TRI == 1 && j <= i) || (TRI == 2 && j < i)) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      double* dA = smem + ((kt + 1) & 1) * 2 * C::OPSZ;
      r2s<WT, false>(dA, ra);
      r2s<WT, false>(dA + C::OPSZ, rb);
    }
    __syncthreads();
  };
  static_assert(WT == 2 || WT == 4, "phase list written for two or four sub-tiles per wave");
  // row sub-tile a = 2 i + wr has non-zeros from step kd = a on
  int kt = 0;
  for (const int end = min(nk, wr); kt < end; ++kt) step(kt, std::integral_constant<int, 0>());
  for (const int end = min(nk, 2 + wr); kt < end; ++kt) step(kt, std::integral_constant<int, 1>());
  if (WT == 4) {
    for (const int end = min(nk, 4 + wr); kt < end; ++kt) step(kt, std::integral_constant<int, 2>());
    for (const int end = min(nk, 6 + wr); kt < end; ++kt) step(kt, std::integral_constant<int, 3>());
  }
  for (; kt < nk; ++kt) step(kt, std::integral_constant<int, WT>());
}

// WT = 4: 128 x 128 tiles; WT = 2: 64 x 64 tiles for launches whose 128-tiles would not fill the device (one or a few emulators: the
// longest tile alone, 125 k-steps at n = 2000, then bounds the kernel)
template <int WT>
__global__ __launch_bounds__(256, 2) void kinv_kernel(BatchView v, int ntiles, int kend) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using C = Cfg<WT>;
  int z, tile;
  decode_block(v.nb, ntiles, z, tile);
  if (z >= v.nb) return;
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tile) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const int emu = slot_to_emu(v.idx, z);
  const int ld = v.LD;
  const double* Li = v.Linv + (size_t)emu * v.MS;
  double* Ki = v.Kinv + (size_t)emu * v.MS;
  const int i0 = ti * C::BM, j0 = tj * C::BM;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  v4d acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) acc[i][j] = (v4d){0., 0., 0., 0.};
  const double* Ag = Li + (size_t)i0 * ld + i0;
  const double* Bg = Li + (size_t)i0 * ld + j0;
  const int nk = (kend - i0) / BK;
  if (nk <= 0) return;
  // (entry (row sub-tile 2 i + wr, column sub-tile 2 j + wc) of the tile; a diagonal tile keeps the pairs b <= a)
  const int tri = (ti != tj) ? 0 : (wr < wc ? 2 : 1);
  if (tri == 0) kinv_mainloop<WT, 0>(Ag, Bg, ld, nk, acc, smem, wr, wc);
  else if (tri == 1) kinv_mainloop<WT, 1>(Ag, Bg, ld, nk, acc, smem, wr, wc);
  else kinv_mainloop<WT, 2>(Ag, Bg, ld, nk, acc, smem, wr, wc);
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      if (tri == 0 || (tri == 1 && j <= i) || (tri == 2 && j < i)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Ki[(size_t)(i0 + (2 * i + wr) * 16 + (lane >> 4) + 4 * r) * ld + j0 + (2 * j + wc) * 16 + (lane & 15)] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// V = Linv * Ks^T written out (NP x MP per slot): first half of the full predictive covariance (fullcov_kernel below).  The
// predictive VARIANCE never stores V (predict_var_w_kernel).  A workgroup takes the pair of row tiles (nti-1-p, p) for its
// column tile, so that every workgroup of the launch is equally long (K = 128 (nti + 1)); 2 x 2 waves, 128 x 128 tiles.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void predict_v_store_kernel(BatchView v, const double* __restrict__ Ks, int MP, int nti, int ntj,
                                                               double* __restrict__ Vout) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using C = Cfg<4>;
  const int npairs = (nti + 1) / 2;
  const int nsr = (npairs + 3) / 4, nsc = (ntj + 15) / 16;
  int z, tile;
  decode_block(v.nb, nsr * nsc * 64, z, tile);
  if (z >= v.nb) return;
  const int st = tile >> 6, w = tile & 63;
  const int pr = (st / nsc) * 4 + (w >> 4), tj = (st % nsc) * 16 + (w & 15);
  if (pr >= npairs || tj >= ntj) return;
  const int emu = slot_to_emu(v.idx, z);
  const int ld = v.LD;
  const double* Li = v.Linv + (size_t)emu * v.MS;
  const double* K = Ks + (size_t)z * MP * ld;
  const int j0 = tj * C::BM;
  const int ti_long = nti - 1 - pr, ti_short = pr;
  double* V = Vout + (size_t)z * v.NP * MP;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && ti_short == ti_long) break;
    const int ti = pass == 0 ? ti_long : ti_short;
    const int i0 = ti * C::BM;
    v4d acc[4][4];
    gemm_mainloop<4, true, true>(Li + (size_t)i0 * ld, ld, K + (size_t)j0 * ld, ld, (i0 + C::BM) / BK, acc, smem);
    for_each_acc<4>(acc, [&](int r, int c, double x) { V[(size_t)(i0 + r) * MP + j0 + c] = x; });
  }
}

// Trailing symmetric update (lower 128 x 128 tiles over rows / columns [c0, NP)) with the 2 x 4-wave main loop:
// the right-looking schedule of a single large matrix spends most of its time here.
__global__ __launch_bounds__(512, 4) void update_tri8_kernel(BatchView v, int c0, int k0, int k1, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using C = WCfg<128, 128, 2, 4>;
  int z, tile;
  decode_block(v.nb, ntiles, z, tile);
  if (z >= v.nb) return;
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tile) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const int emu = slot_to_emu(v.idx, z);
  double* A = v.A + (size_t)emu * v.MS;
  const int ld = v.LD;
  const int i0 = c0 + ti * 128, j0 = c0 + tj * 128;
  v4d acc[C::TI][C::TJ];
  mainloop_w<128, 128, 2, 4>(A + (size_t)i0 * ld + k0, ld, A + (size_t)j0 * ld + k0, ld, (k1 - k0) / BK, acc, smem);
  for_each_acc_w<4>(acc, [&](int r, int c, double x) {
    double* p = A + (size_t)(i0 + r) * ld + (j0 + c);
    *p -= x;
  });
}

// ---------------------------------------------------------------------------------------------
// Predictive-variance kernel, 2 x 4 waves per 128 x 128 block tile (512 threads, 64 x 32 wave tiles, 64 accumulator
// VGPRs -> four waves per SIMD instead of the two of the 2 x 2 / 128-accumulator configuration):
// 59.2 -> 61.1 TFLOP/s on 64 x n=2000 x m=10^4 (4 x 2 waves 60.7, 4 x 4 waves 57.3).
// ---------------------------------------------------------------------------------------------
// (Round 5: the soft lock-step of a super-tile's workgroups -- MOGP_PV_SYNC, persistent workgroups that waited for each other every 8
// k-steps: 3.4 x less L2-miss traffic, 1 - 7 % slower -- and the downward walk of the short row tile, MOGP_PV_DESC, are gone; their
// measurements are in HISTORY.md.)
constexpr bool PV_SWZ = true;      // swizzled ds_read_b128 fragments (mainloop_w<.., SWZ>; round 5: 65.2 -> 66.5 TFLOP/s, profiles/r05_predict_swz_ab.txt)
template <int WR, int WC, bool TRI>
__global__ __launch_bounds__(64 * WR * WC, (WR * WC >= 8 ? 4 : 2)) void predict_var_w_kernel(
    BatchView v, const double* __restrict__ Ks, int MP, int nti, int ntj, double* __restrict__ partial, int lgc, int single) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using C = WCfg<128, 128, WR, WC>;
  // single (launches of fewer than a few rounds of workgroups, e.g. ONE emulator): a workgroup takes one row tile instead of a complementary
  // pair, longest tiles first in dispatch order -- equally long pair tasks that fill the device 1.23 times cost two full rounds (C2,
  // m = 10^4: 632 tasks on 512 slots), unequal single tiles dealt longest-first pack to within a tile of the average
  const int npairs = single ? nti : (nti + 1) / 2;
  const int SC = 1 << lgc, SR = 64 >> lgc;           // super-tile = SR pairs x SC column tiles
  const int nsr = (npairs + SR - 1) / SR, nsc = (ntj + SC - 1) / SC;
  const int t = threadIdx.x, lane = t & 63, wave = TRI ? __builtin_amdgcn_readfirstlane(t >> 6) : (t >> 6);
  const int wr = wave / WC, wc = wave % WC;
  int z, tile;
  decode_block(v.nb, nsr * nsc * 64, z, tile);
  if (z >= v.nb) return;
  const int st = tile >> 6, w = tile & 63;
  const int pr = (st / nsc) * SR + (w >> lgc), tj = (st % nsc) * SC + (w & (SC - 1));
  if (pr >= npairs || tj >= ntj) return;
  const int emu = slot_to_emu(v.idx, z);
  const int ld = v.LD;
  const double* Li = v.Linv + (size_t)emu * v.MS;
  const double* K = Ks + (size_t)z * MP * ld;
  const int j0 = tj * 128;
  const int ti_long = nti - 1 - pr, ti_short = single ? ti_long : pr;
  // both row tiles of a pair walk k upward (free-running workgroups: the order changes neither time nor traffic, round 3)
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && ti_short == ti_long) break;
    const int ti = pass == 0 ? ti_long : ti_short;
    const int i0 = ti * 128;
    v4d acc[C::TI][C::TJ];
    // L^-1 is lower triangular and its rows >= n are padding: TRI skips the structurally zero steps (mainloop_w)
    const int nk = TRI ? min(i0 + 128, (v.n + 15) & ~15) / BK : (i0 + 128) / BK;
    mainloop_w<128, 128, WR, WC, TRI, true, PV_SWZ>(Li + (size_t)i0 * ld, ld, K + (size_t)j0 * ld, ld, nk, acc, smem, i0 / BK, v.n - i0);
    // column sums of squares over the tile's 128 rows: red[wr][128]
    double* red = smem;
#pragma unroll
    for (int j = 0; j < C::TJ; ++j) {
      double s = 0.;
#pragma unroll
      for (int i = 0; i < C::TI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lane < 16) red[wr * 128 + wc * 16 * C::TJ + j * 16 + lane] = s;
    }
    __syncthreads();
    if (t < 128) {
      double s = 0.;
#pragma unroll
      for (int q = 0; q < WR; ++q) s += red[q * 128 + t];
      partial[((size_t)z * nti + ti) * MP + j0 + t] = s;
    }
    __syncthreads();                 // red aliases the operand buffers of the next pass
  }
}

// ---------------------------------------------------------------------------------------------
// full predictive covariance (GaussianProcess.py:899-911):  C = K** - V^T V, V = Linv Ks^T (kend x MP, stored).
// C arrives holding K** (m x m, row stride m); lower 128-tiles are computed and mirrored.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void fullcov_kernel(BatchView v, const double* __restrict__ V, int MP, int m, int ntiles, int kend,
                                                       double* __restrict__ cov) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using C = Cfg<4>;
  int z, tile;
  decode_block(v.nb, ntiles, z, tile);
  if (z >= v.nb) return;
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tile) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const double* Vz = V + (size_t)z * v.NP * MP;
  double* Cz = cov + (size_t)z * m * m;
  const int i0 = ti * C::BM, j0 = tj * C::BM;
  v4d acc[4][4];
  gemm_mainloop<4, false, false>(Vz + i0, MP, Vz + j0, MP, kend / BK, acc, smem);
  for_each_acc<4>(acc, [&](int r, int c, double x) {
    const int i = i0 + r, j = j0 + c;
    if (i < m && j < m) {
      const double val = Cz[(size_t)i * m + j] - x;
      Cz[(size_t)i * m + j] = val;
      if (ti != tj) Cz[(size_t)j * m + i] = val;
    }
  });
}

__global__ void predict_var_finish_kernel(BatchView v, const double* __restrict__ partial, int m, int MP, int nti, double* var, int var_ld) {
  const int z = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const int emu = slot_to_emu(v.idx, z);
  double s = 0.;
  for (int t = 0; t < nti; ++t) s += partial[((size_t)z * nti + t) * MP + j];
  var[(size_t)z * var_ld + j] = v.P[(size_t)emu * v.PS + v.D] - s;
}

// =============================================================================================
// launchers
// =============================================================================================
template <int WT>
static constexpr size_t smem_bytes() { return Cfg<WT>::SMEM_DOUBLES * sizeof(double); }

static int padded_grid(int nb, int ntiles) { return ((nb & 7) == 0) ? nb * ntiles : nb * ntiles; }

void launch_update_narrow(const BatchView& v, int c0, int k0, int k1, hipStream_t s) {
  const int nt = (v.NP - c0) / 64;
  if (nt <= 0) return;
  hipLaunchKernelGGL((update_kernel<2>), dim3(padded_grid(v.nb, nt)), dim3(256), smem_bytes<2>(), s, v, c0, k0, k1, nt, nt);
}

// two adjacent 64-wide block columns [c0, c0+128) in ONE launch (2 nt - 1 lower tiles): twice the workgroups per
// launch, so the last partially filled round of workgroups costs half as much as with two launches
void launch_update_narrow_pair(const BatchView& v, int c0, int k0, int k1, hipStream_t s) {
  const int nt = (v.NP - c0) / 64;
  if (nt <= 0) return;
  const int ntiles = std::max(1, 2 * nt - 1);
  const double m = (double)(v.NP - c0);
  prof_begin("chol_update", s);
  hipLaunchKernelGGL((update_kernel<2>), dim3(padded_grid(v.nb, ntiles)), dim3(256), smem_bytes<2>(), s, v, c0, k0, k1, nt, ntiles);
  // algorithmic flops: the lower part of the m x 128 block column, 2 flops per multiply-add
  prof_end("chol_update", s, (double)v.nb * (m * 128.0 - 128.0 * 128.0 / 2.0) * 2.0 * (k1 - k0), (double)v.nb * (16.0 * m * 128.0 + 8.0 * (m + 128.0) * (k1 - k0)));
}

// one 128-wide block column [c0, c0+128), rows [c0, NP) (inner update of the recursive panel)
void launch_update_wide(const BatchView& v, int c0, int k0, int k1, hipStream_t s) {
  const int nt = (v.NP - c0) / 128;
  if (nt <= 0) return;
  const double m = (double)(v.NP - c0);
  prof_begin("chol_update", s);
  hipLaunchKernelGGL((update_kernel<4>), dim3(padded_grid(v.nb, nt)), dim3(256), smem_bytes<4>(), s, v, c0, k0, k1, nt, nt);
  // algorithmic flops of the block-column update (lower part): (m*128 - 128*128/2) * 2 * K / 2 ... count m*128*K*2 minus the upper half of the diagonal tile
  prof_end("chol_update", s, (double)v.nb * (m * 128.0 - 128.0 * 128.0 / 2.0) * 2.0 * (k1 - k0), (double)v.nb * (16.0 * m * 128.0 + 8.0 * (m + 128.0) * (k1 - k0)));
}

void launch_update_trailing(const BatchView& v, int c0, int k0, int k1, hipStream_t s) {
  const int nt = (v.NP - c0) / 128;
  if (nt <= 0) return;
  const int ntiles = nt * (nt + 1) / 2;
  const double m = (double)(v.NP - c0);
  prof_begin("syrk_trailing", s);
  hipLaunchKernelGGL(update_tri8_kernel, dim3(padded_grid(v.nb, ntiles)), dim3(512), smem_bytes<4>(), s, v, c0, k0, k1, ntiles);
  // algorithmic: lower half of an m x m rank-(k1-k0) update = m^2 (k1-k0) flops; bytes: read+write C lower half + panel
  prof_end("syrk_trailing", s, (double)v.nb * m * m * (k1 - k0), (double)v.nb * (8.0 * m * m + 8.0 * m * (k1 - k0)));
}

void launch_trtri_merges(const BatchView& v, hipStream_t s) {
  // Tile size per level.  Rounds 1 - 3: 64 x 64 tiles at every level (128 x 128 for the upper levels measured slower then: 4.9 vs 4.1 ms
  // at 64 x n=2000 -- a 128-wide tile multiplied the whole diagonal block of its triangular operand).  With the zeros skipped per 16 x 16
  // sub-tile (merge_mainloop) the wide tile no longer pays for them.  Measured, fit + gradient, ms (64 x 64 everywhere / 128 x 128 from
  // h = 1024 / 512 / 256 / 128; before the round: 12.56): 64 x n=2000 12.10 / 11.82 / 11.94 / 11.98 / 11.85; 16 x n=5000 37.83 / 37.30 / 37.51 /
  // 37.44 (38.24); n=16000 70.95 / - / 68.16 / 68.02 (70.85); 8 x n=2000 2.006 / - / 2.083 / 2.094 (2.057): wide tiles where a level has
  // thousands of them.  MOGP_TRTRI_WT4_FROM=<h> forces 128 x 128 tiles from level h on (a huge value: never).
  static const int wt4_from = [] { const char* e = getenv("MOGP_TRTRI_WT4_FROM"); return e ? atoi(e) : -1; }();
  const double alg = ((double)v.n / v.NP) * ((double)v.n / v.NP) * ((double)v.n / v.NP);
  for (int h = 64; h < v.NP; h *= 2) {
    const int nodes = (v.NP + 2 * h - 1) / (2 * h);
    // algorithmic flops of the level: per node two triangular-times-dense products, m2 h^2 + m2^2 h (m2 = rows of the lower block:
    // h, less in the last node when NP is not a power of two); scaled to the n x n matrix (the levels of the padded matrix sum to NP^3 / 3)
    double fl = 0.;
    for (int q = 0; q < nodes; ++q) {
      const double m2 = std::min<double>(h, (double)v.NP - (double)q * 2 * h - h);
      if (m2 > 0) fl += m2 * h * (double)h + m2 * m2 * h;
    }
    if (h > 64) prof_begin("trtri_merge", s);
    const long wide_tiles = (long)v.nb * nodes * (h / 128) * (h / 128);
    if (wt4_from >= 0 ? (h >= wt4_from && h >= 128) : (h >= 512 && wide_tiles >= 3000)) {
      const int tpd = h / 128;
      hipLaunchKernelGGL((trtri_merge_kernel<4, 0>), dim3(padded_grid(v.nb, tpd * tpd * nodes)), dim3(256), smem_bytes<4>(), s, v, h, tpd, nodes);
      hipLaunchKernelGGL((trtri_merge_kernel<4, 1>), dim3(padded_grid(v.nb, tpd * tpd * nodes)), dim3(256), smem_bytes<4>(), s, v, h, tpd, nodes);
    } else {
      const int tpd = h / 64;
      hipLaunchKernelGGL((trtri_merge_kernel<2, 0>), dim3(padded_grid(v.nb, tpd * tpd * nodes)), dim3(256), smem_bytes<2>(), s, v, h, tpd, nodes);
      hipLaunchKernelGGL((trtri_merge_kernel<2, 1>), dim3(padded_grid(v.nb, tpd * tpd * nodes)), dim3(256), smem_bytes<2>(), s, v, h, tpd, nodes);
    }
    if (h > 64) prof_end("trtri_merge", s, (double)v.nb * fl * alg, 0.);
  }
}

// predict_var_q_kernel (round 5): the same task -- a pair of 128-row tiles of L^-1 against one column tile of K*, column sums of squares -- on
// the k-step of the one-launch Cholesky's GEMM tasks (mainloop_qt: three swizzled LDS stages, the next fragments requested before the barrier,
// the step's instruction order prescribed).  256 threads (2 x 2 waves) per 128 x 64 tile, two workgroups per CU: alone in the probe that loop
// reaches 0.91 of the fp64 MFMA peak with two waves per SIMD, where the 8-wave 128 x 128 form above (four waves per SIMD, 128 registers, two
// stages, fragments read in front of the MFMAs that need them) stands at 0.83 - 0.85.  Column tiles are 64 points wide: K* is read as often as
// before (half as wide a panel, twice as many of them), L^-1 twice as often.
template <bool TRI>
__global__ __launch_bounds__(256, 2) void predict_var_q_kernel(BatchView v, const double* __restrict__ Ks, int MP, int nti, int ntj,
                                                               double* __restrict__ partial, int lgc, int single) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using C = WCfg<128, 64, 2, 2>;
  const int npairs = single ? nti : (nti + 1) / 2;
  const int SC = 1 << lgc, SR = 64 >> lgc;           // super-tile = SR pairs x SC column tiles
  const int nsr = (npairs + SR - 1) / SR, nsc = (ntj + SC - 1) / SC;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave / 2, wc = wave % 2;
  int z, tile;
  decode_block(v.nb, nsr * nsc * 64, z, tile);
  if (z >= v.nb) return;
  const int st = tile >> 6, w = tile & 63;
  const int pr = (st / nsc) * SR + (w >> lgc), tj = (st % nsc) * SC + (w & (SC - 1));
  if (pr >= npairs || tj >= ntj) return;
  const int emu = slot_to_emu(v.idx, z);
  const int ld = v.LD;
  const double* Li = v.Linv + (size_t)emu * v.MS;
  const double* K = Ks + (size_t)z * MP * ld;
  const int j0 = tj * 64;
  const int ti_long = nti - 1 - pr, ti_short = single ? ti_long : pr;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && ti_short == ti_long) break;
    const int ti = pass == 0 ? ti_long : ti_short;
    const int i0 = ti * 128;
    v4d acc[C::TI][C::TJ];
    const int nk = min(i0 + 128, (v.n + 15) & ~15) / BK;
    mainloop_qt<128, 64, 2, 2>(Li + (size_t)i0 * ld, ld, K + (size_t)j0 * ld, ld, nk, acc, smem, TRI ? i0 / BK : nk, TRI ? v.n - i0 : 128);
    // column sums of squares over the tile's 128 rows: red[wr][64]
    double* red = smem;
#pragma unroll
    for (int j = 0; j < C::TJ; ++j) {
      double s = 0.;
#pragma unroll
      for (int i = 0; i < C::TI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lane < 16) red[wr * 64 + wc * 16 * C::TJ + j * 16 + lane] = s;
    }
    __syncthreads();
    if (t < 64) partial[((size_t)z * nti + ti) * MP + j0 + t] = red[t] + red[64 + t];
    __syncthreads();                 // red aliases the operand buffers of the next pass
  }
}

void launch_kinv(const BatchView& v, int n_cu, hipStream_t s) {
  const int kend = ((v.n + 15) / 16) * 16;
  const int nt = (v.n + 127) / 128;      // tiles that contain real rows
  const int ntiles = nt * (nt + 1) / 2;
  // 64 x 64 tiles when the 128 x 128 ones would fill the device less than twice (MOGP_KINV_WT = 2 / 4 forces either)
  static const int force_wt = [] { const char* e = getenv("MOGP_KINV_WT"); return e ? atoi(e) : 0; }();
  const bool small = force_wt ? force_wt == 2 : ((long)v.nb * ntiles < 2L * 2 * n_cu);
  prof_begin("kinv", s);
  if (small) {
    const int nt2 = (v.n + 63) / 64;
    const int ntiles2 = nt2 * (nt2 + 1) / 2;
    hipLaunchKernelGGL(kinv_kernel<2>, dim3(padded_grid(v.nb, ntiles2)), dim3(256), smem_bytes<2>(), s, v, ntiles2, kend);
  } else
    hipLaunchKernelGGL(kinv_kernel<4>, dim3(padded_grid(v.nb, ntiles)), dim3(256), smem_bytes<4>(), s, v, ntiles, kend);
  prof_end("kinv", s, (double)v.nb * (double)v.n * v.n * v.n / 3.0, 0.);
}

void launch_predict_var(const BatchView& v, const double* Ks, int m, int MP, double* partial, double* var, int var_ld, int n_cu, hipStream_t s) {
  const int nti = (v.n + 127) / 128, ntj = MP / 128;
  prof_begin("predict_var", s);
  // 2 x 4 waves per 128 x 128 tile (measured, TFLOP/s, dense form: 2 x 2 waves 59.2, 2 x 4 waves 61.1, 4 x 2 waves 60.7, 4 x 4 waves 57.3).
  // super-tile = 2^lgc column tiles x 64/2^lgc row-tile pairs.  Measured at nti = 16 (8 pairs), m = 5632, L2-miss bytes per launch /
  // TFLOP/s: 8x8 37 GB / 62.3, 4 pairs x 16 44 GB / 61.9, 2 x 32 53 GB / 60.4, 1 x 64 54 GB / 60.4; without the XCD-aware
  // block decode (workgroups of a super-tile spread over all eight L2s) 47 GB but only 52.1 TFLOP/s
  static const int lgc_env = [] { const char* e = getenv("MOGP_PV_LGC"); return e ? atoi(e) : -1; }();
  const int lgc = lgc_env >= 0 ? lgc_env : 3;
  // MOGP_PV_SINGLE = 0 / 1 forces pairs / single row tiles; default: single row tiles when the pair tasks fill the device fewer than twice.
  // Measured (predict incl. host copies, m = 10^4, ms, pairs / single): 1 x n=2000 1.080 / 1.031, 2 x 1.82 / 1.84, 3 x 2.44 / 2.50, 4 x 3.05 /
  // 3.27, 1 x n=5000 5.36 / 4.86, 1 x n=700 (m = 3000) 0.213 / 0.187; bit-identical
  static const int force_single = [] { const char* e = getenv("MOGP_PV_SINGLE"); return e ? atoi(e) : -1; }();
  {
    const int SC = 1 << lgc, SR = 64 >> lgc;
    const int nst = (((nti + 1) / 2 + SR - 1) / SR) * ((ntj + SC - 1) / SC);
    const long pair_tasks = (long)v.nb * ((nti + 1) / 2) * ntj;
    const bool single = force_single >= 0 ? force_single != 0 : pair_tasks < 2L * 2 * n_cu;
    // predict_var_q_kernel for launches of single row tiles (one n = 2000 matrix, 10^4 points: 0.82 -> 0.70 ms); MOGP_PV_Q = 0 / 1 forces either
    // kernel.  For full launches the two are level -- 64 x n=2000 66.5 / 66.4 TFLOP/s, n=16000 68.2 / 68.2, 16 x n=5000 66.3 / 65.3 -- two kernels
    // that share nothing but the MFMA instruction end at the same rate: the part is at its power limit there (1.31 kW, 2.28 - 2.30 GHz under
    // either; profiles/r05_predict_q_ab.txt)
    static const int force_q = [] { const char* e = getenv("MOGP_PV_Q"); return e ? atoi(e) : -1; }();
    const bool use_q = force_q >= 0 ? force_q != 0 : single;
    if (use_q) {
      // (64-point column tiles: sixteen of them per super-tile -- the 1024 points of eight 128-point tiles -- for single row tiles: one matrix
      // 52.5 -> 57.1 TFLOP/s; full launches are fastest with eight)
      const int lgq = lgc_env >= 0 ? lgc_env : (single ? 4 : 3);
      const int SC = 1 << lgq, SR = 64 >> lgq;
      const int ntq = MP / 64;
      constexpr size_t lds_q = (size_t)QCfg<128, 64>::SMEM_DOUBLES * sizeof(double);
      const int nsq = ((((single ? nti : (nti + 1) / 2)) + SR - 1) / SR) * ((ntq + SC - 1) / SC);
      hipLaunchKernelGGL((predict_var_q_kernel<true>), dim3(padded_grid(v.nb, nsq * 64)), dim3(256), lds_q, s, v, Ks, MP,
                         nti, ntq, partial, lgq, single ? 1 : 0);
    } else if (single) {
      const int nst1 = ((nti + SR - 1) / SR) * ((ntj + SC - 1) / SC);
      hipLaunchKernelGGL((predict_var_w_kernel<2, 4, true>), dim3(padded_grid(v.nb, nst1 * 64)), dim3(512), smem_bytes<4>(), s, v, Ks, MP, nti, ntj, partial, lgc, 1);
    } else
      hipLaunchKernelGGL((predict_var_w_kernel<2, 4, true>), dim3(padded_grid(v.nb, nst * 64)), dim3(512), smem_bytes<4>(), s, v, Ks, MP, nti, ntj, partial, lgc, 0);
  }
  prof_end("predict_var", s, (double)v.nb * (double)m * v.n * v.n, 0.);
  hipLaunchKernelGGL(predict_var_finish_kernel, dim3((m + 255) / 256, v.nb), dim3(256), 0, s, v, partial, m, MP, nti, var, var_ld);
}

void launch_trsm128(const BatchView& v, int c0, const double* Lpack128, hipStream_t s) {
  const int rows = v.NP - c0 - 128;
  if (rows <= 0) return;
  prof_begin("chol_trsm128", s);
  hipLaunchKernelGGL(trsm128_lds_kernel, dim3(rows / 64, v.nb), dim3(256), TRSM128L_LDS * sizeof(double), s, v, c0, c0 + 128, Lpack128);
  // rows x 128 triangular solve: rows * 128^2 flops; the panel is read and written once
  prof_end("chol_trsm128", s, (double)v.nb * rows * 128.0 * 128.0, (double)v.nb * 2.0 * 8.0 * rows * 128.0);
}

// cov (nb, m, m) holds K** on entry and the predictive covariance (without nugget) on return; V is nb*NP*MP scratch
void launch_predict_fullcov(const BatchView& v, const double* Ks, int m, int MP, double* V, double* cov, hipStream_t s) {
  const int nti = (v.n + 127) / 128, ntj = MP / 128;
  const int nsup = (((nti + 1) / 2 + 3) / 4) * ((ntj + 15) / 16) * 64;
  prof_begin("predict_var", s);
  hipLaunchKernelGGL(predict_v_store_kernel, dim3(padded_grid(v.nb, nsup)), dim3(256), smem_bytes<4>(), s, v, Ks, MP, nti, ntj, V);
  prof_end("predict_var", s, (double)v.nb * (double)m * v.n * v.n, 0.);
  const int ntiles = ntj * (ntj + 1) / 2;
  const int kend = ((v.n + 15) / 16) * 16;
  prof_begin("fullcov_syrk", s);
  hipLaunchKernelGGL(fullcov_kernel, dim3(padded_grid(v.nb, ntiles)), dim3(256), smem_bytes<4>(), s, v, V, MP, m, ntiles, kend, cov);
  prof_end("fullcov_syrk", s, (double)v.nb * (double)m * m * v.n, 0.);
}

}  // namespace mogp
