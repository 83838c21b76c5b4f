// Device-side MFMA main loop shared by the tile kernels (kernels_gemm.hip) and the one-launch Cholesky (kernels_mchol.hip).
// fp64 C/D fragment map of v_mfma_f64_16x16x4_f64 (verified on hardware by tools/mfma_probe.hip):
//   row = (lane>>4) + 4*reg, col = lane&15;  A: A[lane&15][lane>>4];  B: B[lane>>4][lane&15].
#pragma once
#include <type_traits>
#include "launch.h"

namespace mogp {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
constexpr int LDK = BK + 2;
// XOR swizzle of the 16-byte chunk index of an UNPADDED [row][16] operand tile (mainloop_q, mainloop_w<.., SWZ>): chunk c of row R lies at
// c ^ q_swz(R & 15), which makes the ds_read_b128 fragment reads conflict-free in the four 16-lane groups the instruction is served in
__device__ __forceinline__ int q_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }

// ---------------------------------------------------------------------------------------------
// Main loop with WR x WC waves per BM x BN block tile, both operands K-major (same LDS layout and k-step as
// gemm_mainloop).  More, smaller wave tiles than the 2 x 2 configuration: fewer accumulator VGPRs per wave, so
// more waves per SIMD are resident and more of the LDS / barrier latency is covered.
//   acc[i][j]: MFMA tile rows wr*16*TI + 16 i, columns wc*16*TJ + 16 j  (TI = BM/16/WR, TJ = BN/16/WC)
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WR, int WC>
struct WCfg {
  static constexpr int NT = 64 * WR * WC;               // threads
  static constexpr int TI = BM / 16 / WR, TJ = BN / 16 / WC;
  static constexpr int CHA = BM * 8 / NT, CHB = BN * 8 / NT;   // 16-byte chunks per thread per operand tile
  static constexpr int OPA = BM * LDK, OPB = BN * LDK;
  static constexpr int SMEM_DOUBLES = 2 * (OPA + OPB);
  static_assert(CHA >= 1 && CHB >= 1, "operand tile smaller than one chunk per thread");
};

// TRIA: the A operand is a row tile of a LOWER-TRIANGULAR matrix whose last 128 columns of the k range are its diagonal
// block (nk_full = 16-deep k-steps before it), and rows >= a_rows of the tile are padding.  The 16-row MFMA sub-tiles are
// then dealt to the WR wave rows round-robin (sub-tile a = i * WR + wr: every wave owns sub-tiles from the top and the
// bottom of the tile) and a wave skips what is structurally zero:
//   * step kd of the diagonal block only has non-zeros in sub-tiles a >= kd  (36 of the 64 (sub-tile, step) pairs);
//   * sub-tiles that start at or below row a_rows are padding.
// The skipped products are exact zeros.  Control flow: the k loop is cut into consecutive loops, one per set of active
// sub-tiles [S, E) -- each one is the dense straight-line step restricted to those sub-tiles, the accumulators of the
// others are simply not touched -- followed by a loop that only moves the wave's share of the operand tiles.  (A branch
// around MFMA groups inside one loop costs more in register copies and exposed LDS latency than the skipped MFMAs save.)
// Callers must not depend on the row -> accumulator map.
// ZERO = false: the products are ADDED to what acc holds on entry (a k range continued after a wait).
// SWZ (round 5, the predictive variance): operand tiles unpadded and swizzled as in mainloop_q, fragments as ds_read_b128 -- half the LDS
// instructions and no bank conflicts (the padded [row][18] layout spends 40 % of its LDS cycles in conflicts, profiles/r05_loop_probe_pmc.txt);
// lane (fr, fk) then multiplies k = 4 fk + 2 h + e in the MFMA (h, e) of a step (results agree with the padded form to rounding).
template <int BM, int BN, int WR, int WC, bool TRIA = false, bool ZERO = true, bool SWZ = false>
__device__ __forceinline__ void mainloop_w(const double* __restrict__ Ag, int lda, const double* __restrict__ Bg, int ldb, int nk,
                                           v4d (&acc)[WCfg<BM, BN, WR, WC>::TI][WCfg<BM, BN, WR, WC>::TJ], double* smem,
                                           int nk_full = 0, int a_rows = BM) {
  using C = WCfg<BM, BN, WR, WC>;
  const int t = mogp_tid(), lane = t & 63;
  const int wave = TRIA ? __builtin_amdgcn_readfirstlane(t >> 6) : (t >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int fr = lane & 15, fk = lane >> 4;
  if (ZERO) {
#pragma unroll
    for (int i = 0; i < C::TI; ++i)
#pragma unroll
      for (int j = 0; j < C::TJ; ++j) acc[i][j] = (v4d){0., 0., 0., 0.};
  }
  if (nk <= 0) return;
  v2d ra[C::CHA], rb[C::CHB];
  // chunk q of a thread = rows (t >> 3) + q NT/8: ONE 32-bit byte offset per operand and thread against wave-uniform bases (scalar
  // base + vector offset addressing: no 64-bit address arithmetic per load, three address registers fewer than per-thread pointers)
  const unsigned offA = (unsigned)(((t >> 3) * lda + (t & 7) * 2) * (int)sizeof(double));
  const unsigned offB = (unsigned)(((t >> 3) * ldb + (t & 7) * 2) * (int)sizeof(double));
  auto loadA = [&]() {
#pragma unroll
    for (int q = 0; q < C::CHA; ++q)
      ra[q] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Ag + (size_t)q * (C::NT / 8) * lda) + offA);
  };
  auto loadB = [&]() {
#pragma unroll
    for (int q = 0; q < C::CHB; ++q)
      rb[q] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Bg + (size_t)q * (C::NT / 8) * ldb) + offB);
  };
  // (SWZ: row (t >> 3) + (NT / 8) q, chunk (t & 7) ^ sw(row) -- NT / 8 is a multiple of 16, so the swizzle is the thread's own)
  const int st_swz = (t >> 3) * BK + (((t & 7) ^ q_swz((t >> 3) & 15)) << 1);
  auto store = [&](double* sA, double* sB) {
#pragma unroll
    for (int q = 0; q < C::CHA; ++q) {
      const int c = t + C::NT * q;
      if (SWZ) *reinterpret_cast<v2d*>(sA + st_swz + q * (C::NT / 8) * BK) = ra[q];
      else *reinterpret_cast<v2d*>(sA + (c >> 3) * LDK + (c & 7) * 2) = ra[q];
    }
#pragma unroll
    for (int q = 0; q < C::CHB; ++q) {
      const int c = t + C::NT * q;
      if (SWZ) *reinterpret_cast<v2d*>(sB + st_swz + q * (C::NT / 8) * BK) = rb[q];
      else *reinterpret_cast<v2d*>(sB + (c >> 3) * LDK + (c & 7) * 2) = rb[q];
    }
  };
  loadA();
  loadB();
  store(smem, smem + C::OPA);
  __syncthreads();
  // one k-step with the sub-tiles [S, E) of this wave
  auto step = [&](int kt, auto S_, auto E_) {
    constexpr int S = decltype(S_)::value, E = decltype(E_)::value;
    const int par = kt & 1;
    const double* sA = smem + par * (C::OPA + C::OPB);
    const double* sB = sA + C::OPA;
    const bool more = kt + 1 < nk;
    if (more) {
      Ag += BK;
      Bg += BK;
      loadA();
      loadB();
    }
    if (S < E && SWZ) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v2d a2[C::TI], b2[C::TJ];
        const int fo = fr * BK + ((((2 * fk) ^ q_swz(fr)) << 1) ^ (h << 1));
#pragma unroll
        for (int i = S; i < E; ++i) a2[i] = *reinterpret_cast<const v2d*>(sA + (TRIA ? i * WR + wr : wr * C::TI + i) * 16 * BK + fo);
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) b2[j] = *reinterpret_cast<const v2d*>(sB + (wc * C::TJ + j) * 16 * BK + fo);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int i = S; i < E; ++i)
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[i][e], b2[j][e], acc[i][j], 0, 0, 0);
      }
    } else if (S < E) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double a[C::TI], b[C::TJ];
        const int k = kk * 4 + fk;
#pragma unroll
        for (int i = S; i < E; ++i) a[i] = sA[((TRIA ? i * WR + wr : wr * C::TI + i) * 16 + fr) * LDK + k];
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) b[j] = sB[(wc * 16 * C::TJ + j * 16 + fr) * LDK + k];
#pragma unroll
        for (int i = S; i < E; ++i)
#pragma unroll
          for (int j = 0; j < C::TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      double* dA = smem + (par ^ 1) * (C::OPA + C::OPB);
      store(dA, dA + C::OPA);
    }
    __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>;
  using IT = std::integral_constant<int, C::TI>;
  int kt = 0;
  if (!TRIA) {
    for (; kt < nk; ++kt) step(kt, I0(), IT());
    return;
  }
  static_assert(!TRIA || C::TI == 4, "the phase dispatch below is written for four sub-tiles per wave");
  // sub-tile i (a = i WR + wr) has non-zeros up to step a of the diagonal block: phase S lasts while sub-tile S is active
  auto phases = [&](auto E_) {
    constexpr int E = decltype(E_)::value;
    if (E > 0) for (const int end = min(nk, nk_full + 0 * WR + wr + 1); kt < end; ++kt) step(kt, std::integral_constant<int, 0>(), E_);
    if (E > 1) for (const int end = min(nk, nk_full + 1 * WR + wr + 1); kt < end; ++kt) step(kt, std::integral_constant<int, 1>(), E_);
    if (E > 2) for (const int end = min(nk, nk_full + 2 * WR + wr + 1); kt < end; ++kt) step(kt, std::integral_constant<int, 2>(), E_);
    if (E > 3) for (const int end = min(nk, nk_full + 3 * WR + wr + 1); kt < end; ++kt) step(kt, std::integral_constant<int, 3>(), E_);
  };
  int n_act = 0;     // sub-tiles of this wave that contain real rows
#pragma unroll
  for (int i = 0; i < C::TI; ++i) n_act += ((i * WR + wr) * 16 < a_rows) ? 1 : 0;
  if (n_act == 4) phases(std::integral_constant<int, 4>());
  else if (n_act == 3) phases(std::integral_constant<int, 3>());
  else if (n_act == 2) phases(std::integral_constant<int, 2>());
  else if (n_act == 1) phases(std::integral_constant<int, 1>());
  for (; kt < nk; ++kt) step(kt, I0(), I0());
}

// Same k-step as mainloop_w (both operands K-major, LDS double buffer, one barrier per 16-deep step), but the global
// loads run PD steps ahead of the matrix cores through a ring of PD register sets, and the products are ADDED to acc.
// Written for the one-launch Cholesky: the last 128 columns of a tile's k range are read straight after another
// workgroup has published them write-through, i.e. from HBM / the infinity cache, and with the one-step prefetch of
// mainloop_w every one of its 8 steps exposed a full memory latency (25-35 us for 8 steps, per-task stamps of
// tools/mchol_trace.py); four steps ahead the latency is paid about twice.  nk must be a multiple of PD.
// (Rounds 2 - 4 let a GEMM task PARK -- sleep at a barrier while a diagonal-block task ran in the co-resident workgroup of its CU; on the
// round-5 kernels parking never wins, profiles/r05_regime_sweep.txt, and is gone.)
template <int BM, int BN, int WR, int WC, int PD>
__device__ __forceinline__ void mainloop_pf(const double* __restrict__ Ag, int lda, const double* __restrict__ Bg, int ldb, int nk,
                                            v4d (&acc)[WCfg<BM, BN, WR, WC>::TI][WCfg<BM, BN, WR, WC>::TJ], double* smem, int kmask = -1) {
  // kmask (measurement only, MOGP_MC_NOTRAFFIC): k-step kt reads the operand columns of step kt & kmask -- with kmask = 3 every task
  // re-reads its first 64 columns from the caches: the same instruction stream without the memory traffic (results are garbage)
  using C = WCfg<BM, BN, WR, WC>;
  const int t = mogp_tid(), lane = t & 63, wave = t >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int fr = lane & 15, fk = lane >> 4;
  if (nk <= 0) return;
  v2d ra[PD][C::CHA], rb[PD][C::CHB];
  // one 32-bit byte offset per operand and thread against wave-uniform bases (as in mainloop_w)
  const unsigned offA = (unsigned)(((t >> 3) * lda + (t & 7) * 2) * (int)sizeof(double));
  const unsigned offB = (unsigned)(((t >> 3) * ldb + (t & 7) * 2) * (int)sizeof(double));
  auto load = [&](int u, int kt_) {
    const int kt = kt_ & kmask;
#pragma unroll
    for (int q = 0; q < C::CHA; ++q)
      ra[u][q] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Ag + (size_t)q * (C::NT / 8) * lda + (size_t)kt * BK) + offA);
#pragma unroll
    for (int q = 0; q < C::CHB; ++q)
      rb[u][q] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Bg + (size_t)q * (C::NT / 8) * ldb + (size_t)kt * BK) + offB);
  };
  auto store = [&](int u, double* sA, double* sB) {
#pragma unroll
    for (int q = 0; q < C::CHA; ++q) {
      const int c = t + C::NT * q;
      *reinterpret_cast<v2d*>(sA + (c >> 3) * LDK + (c & 7) * 2) = ra[u][q];
    }
#pragma unroll
    for (int q = 0; q < C::CHB; ++q) {
      const int c = t + C::NT * q;
      *reinterpret_cast<v2d*>(sB + (c >> 3) * LDK + (c & 7) * 2) = rb[u][q];
    }
  };
#pragma unroll
  for (int u = 0; u < PD; ++u) load(u, u);
  store(0, smem, smem + C::OPA);
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int kt = kt0 + u;
      const double* sA = smem + (kt & 1) * (C::OPA + C::OPB);
      const double* sB = sA + C::OPA;
      if (kt + PD < nk) load(u, kt + PD);                 // register set u held step kt, which is in LDS already
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double a[C::TI], b[C::TJ];
        const int k = kk * 4 + fk;
#pragma unroll
        for (int i = 0; i < C::TI; ++i) a[i] = sA[((wr * C::TI + i) * 16 + fr) * LDK + k];
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) b[j] = sB[(wc * 16 * C::TJ + j * 16 + fr) * LDK + k];
#pragma unroll
        for (int i = 0; i < C::TI; ++i)
#pragma unroll
          for (int j = 0; j < C::TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (kt + 1 < nk) {
        double* dA = smem + ((kt + 1) & 1) * (C::OPA + C::OPB);
        store((u + 1) % PD, dA, dA + C::OPA);
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// mainloop_q (round 5): the k-step of mainloop_pf rebuilt so that ONE wave per SIMD keeps the matrix pipe busy -- in the one-launch
// Cholesky a GEMM task runs alone on its CU's pipes whenever the co-resident workgroup is in a solve / load / wait phase (a third of
// the time), and alone mainloop_pf reaches 63 - 72 % of the pipe (k-step 1.35 us against 0.85 - 0.97).  Three changes:
//   * operand tiles UNPADDED [row][16] in LDS with the 16-byte chunk index XOR-swizzled by the row (chunk c of row R at
//     c ^ sw(R & 15), sw(r) = ((r >> 1) & 1) | ((r >> 3) & 1) << 2): a lane's two k values of a half step are ONE ds_read_b128,
//     conflict-free in the four 16-lane groups the instruction is served in (12 reads per k-step instead of 24 ds_read_b64, and
//     ds_read_b128 reaches its rate from one wave per SIMD, ds_read_b64 only from four -- MI355X_MICROARCH.md, LDS);
//     lane (fr, fk) therefore multiplies k = 4 fk + 2 h + e in the MFMA (h, e) of a step: the k order INSIDE a 16-deep step differs
//     from mainloop_pf's (results agree to rounding, not bit for bit);
//   * THREE LDS stages, one barrier per step: the stage a step writes was last read a whole step ago and is first read a whole step
//     later, so neither the LDS writes nor the first fragment reads of a step sit next to the barrier -- the fragments of half step 0
//     of step kt + 1 are requested BEFORE the barrier that ends step kt;
//   * global loads G steps ahead of the LDS stage they fill (2 + G steps ahead of the matrix cores) through G register sets --
//     with G = 2 the same four steps as mainloop_pf<.., 4> with half the staging registers.
// The products are ADDED to acc.  nk: a positive multiple of G.  LDS: 3 * (BM + BN) * 16 doubles.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN>
struct QCfg {
  static constexpr int STAGE = (BM + BN) * BK;           // doubles per stage: A tile, then B tile
  static constexpr int SMEM_DOUBLES = 3 * STAGE;
};

// Instruction order of a step for the scheduler (SCHED = 2): one memory operation behind each of the first MFMAs of a half step, so that
// LDS reads / writes and global loads ISSUE while the matrix pipe works -- left alone the compiler puts the step's 12 reads and 6 writes in
// front of its first MFMA and waits for all of them there (one wave per SIMD: nobody else fills the pipe meanwhile).
template <int MASK, int SIZE>
__device__ __forceinline__ void sched_group() { __builtin_amdgcn_sched_group_barrier(MASK, SIZE, 0); }
template <int K, int HI, int NR, int NW>
__device__ __forceinline__ void sched_ops() {        // memory operations K .. HI-1 of a half step: NR LDS reads, NW LDS writes, then global loads
  if constexpr (K < HI) {
    if constexpr (K < NR) sched_group<0x100, 1>();
    else if constexpr (K < NR + NW) sched_group<0x200, 1>();
    else sched_group<0x020, 1>();
    sched_ops<K + 1, HI, NR, NW>();
  }
}
template <int M, int NM, int NR, int NW, int NL>
__device__ __forceinline__ void sched_half() {       // MFMA M of NM, then its share of the NR + NW + NL memory operations
  if constexpr (M < NM) {
    sched_group<0x008, 1>();
    constexpr int TOT = NR + NW + NL;
    sched_ops<M * TOT / NM, (M + 1) * TOT / NM, NR, NW>();
    sched_half<M + 1, NM, NR, NW, NL>();
  }
}

// kmask: as in mainloop_pf.  SCHED: 0 the compiler's order, 1 the barrier pinned behind the step's last MFMAs, 2 also the interleave above.
template <int BM, int BN, int WR, int WC, int G, int SCHED = 0>
__device__ __forceinline__ void mainloop_q(const double* __restrict__ Ag, int lda, const double* __restrict__ Bg, int ldb, int nk,
                                           v4d (&acc)[WCfg<BM, BN, WR, WC>::TI][WCfg<BM, BN, WR, WC>::TJ], double* smem, int kmask = -1) {
  using C = WCfg<BM, BN, WR, WC>;
  constexpr int STAGE = QCfg<BM, BN>::STAGE;
  const int t = mogp_tid(), lane = t & 63, wave = t >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int fr = lane & 15, fk = lane >> 4;
  if (nk <= 0) return;
  v2d ra[G][C::CHA], rb[G][C::CHB];
  const unsigned offA = (unsigned)(((t >> 3) * lda + (t & 7) * 2) * (int)sizeof(double));
  const unsigned offB = (unsigned)(((t >> 3) * ldb + (t & 7) * 2) * (int)sizeof(double));
  auto load = [&](int u, int kt_) {
    const int kt = kt_ & kmask;
#pragma unroll
    for (int q = 0; q < C::CHA; ++q)
      ra[u][q] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Ag + (size_t)q * (C::NT / 8) * lda + (size_t)kt * BK) + offA);
#pragma unroll
    for (int q = 0; q < C::CHB; ++q)
      rb[u][q] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Bg + (size_t)q * (C::NT / 8) * ldb + (size_t)kt * BK) + offB);
  };
  // this thread's chunk of a staged row: row (t >> 3) + 32 q, 16-byte chunk (t & 7) ^ sw(row)
  const int st_off = (t >> 3) * BK + (((t & 7) ^ q_swz((t >> 3) & 15)) << 1);
  auto store = [&](int u, double* st) {
#pragma unroll
    for (int q = 0; q < C::CHA; ++q) *reinterpret_cast<v2d*>(st + st_off + q * (C::NT / 8) * BK) = ra[u][q];
#pragma unroll
    for (int q = 0; q < C::CHB; ++q) *reinterpret_cast<v2d*>(st + BM * BK + st_off + q * (C::NT / 8) * BK) = rb[u][q];
  };
  // fragment of half step h: the lane's k = 4 fk + 2 h, + 1 of row fr of a 16-row block
  const int fo0 = fr * BK + (((2 * fk) ^ q_swz(fr)) << 1);
  v2d fa[2][C::TI], fb[2][C::TJ];
  auto frag = [&](int set, const double* st, int h) {
    const int fo = fo0 ^ (h << 1);
#pragma unroll
    for (int i = 0; i < C::TI; ++i) fa[set][i] = *reinterpret_cast<const v2d*>(st + (wr * C::TI + i) * 16 * BK + fo);
#pragma unroll
    for (int j = 0; j < C::TJ; ++j) fb[set][j] = *reinterpret_cast<const v2d*>(st + BM * BK + (wc * C::TJ + j) * 16 * BK + fo);
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < C::TI; ++i)
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][i][e], fb[set][j][e], acc[i][j], 0, 0, 0);
  };
  // Branch-free body: loads, stores and fragment reads past the end of the k range are made harmless instead of skipped (a load
  // re-reads the last step, a store fills a stage nobody reads any more) -- with conditions around them the compiler splits the step into
  // a dozen basic blocks and serialises what should overlap.  nk must be a multiple of G.
  const int last = nk - 1;
  // prologue: stages 0 and 1 filled, register sets hold steps 2 .. 2 + G - 1
  load(0, 0);
  load(1 % G, min(1, last));
  store(0, smem);
  store(1 % G, smem + STAGE);
#pragma unroll
  for (int u = 0; u < G; ++u) load(u, min(2 + u, last));
  __syncthreads();
  frag(0, smem, 0);
  int cur = 0;                                            // stage of step kt
  for (int kt0 = 0; kt0 < nk; kt0 += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int kt = kt0 + u;
      const double* st = smem + cur * STAGE;
      const int nxt = cur == 2 ? 0 : cur + 1, wrt = nxt == 2 ? 0 : nxt + 1;
      frag(1, st, 1);
      mfmas(0);
      // the stage of step kt + 2 (last read during step kt - 1, a barrier ago) gets register set u; the set is then reloaded
      store(u, smem + wrt * STAGE);
      load(u, min(kt + 2 + G, last));
      frag(0, smem + nxt * STAGE, 0);                      // written during step kt - 1: visible since the barrier that ended it
      mfmas(1);
      if (SCHED == 2) {
        constexpr int NM = 2 * C::TI * C::TJ, NR = C::TI + C::TJ, NW = C::CHA + C::CHB;
        sched_half<0, NM, NR, NW, NW>();                   // first half step: frag(1) reads, the stage's writes, the next global loads
        sched_half<0, NM, NR, 0, 0>();                     // second: the first fragments of the next step
      }
      if (SCHED == 3) {                                    // (probe: the next step's fragments early in the second half, two behind each MFMA)
        constexpr int NM = 2 * C::TI * C::TJ, NR = C::TI + C::TJ, NW = C::CHA + C::CHB;
        sched_half<0, NM, NR, NW, NW>();
        sched_half<0, NR / 2, NR, 0, 0>();
        sched_group<0x008, NM - NR / 2>();
      }
      if (SCHED >= 1) __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise lifts the barrier -- and its wait for the reads just requested -- above these MFMAs)
      __syncthreads();
      // (tried: the barrier in the MIDDLE of the step, so that every LDS operation in flight at it was issued half a step earlier -- the
      // same times, tools/gemm_loop_probe.hip round 5)
      cur = nxt;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// mainloop_qt (round 5, the predictive variance): the k-step of mainloop_q -- three swizzled LDS stages, fragments of the next half step requested
// before the barrier, global loads two steps ahead of their stage, the step's instruction order prescribed (SCHED = 2) -- for a row tile of a
// LOWER-TRIANGULAR A operand (TRIA of mainloop_w: sub-tile a = i * WR + wr, structurally zero (sub-tile, step) pairs of the diagonal block and
// padding sub-tiles skipped through consecutive loops with fixed active sets [S, E)).  acc starts at zero.  LDS: QCfg<BM, BN>::SMEM_DOUBLES.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WR, int WC>
__device__ __forceinline__ void mainloop_qt(const double* __restrict__ Ag, int lda, const double* __restrict__ Bg, int ldb, int nk,
                                            v4d (&acc)[WCfg<BM, BN, WR, WC>::TI][WCfg<BM, BN, WR, WC>::TJ], double* smem, int nk_full, int a_rows) {
  using C = WCfg<BM, BN, WR, WC>;
  constexpr int STAGE = QCfg<BM, BN>::STAGE;
  static_assert(C::TI == 4, "the phase dispatch below is written for four sub-tiles per wave");
  const int t = mogp_tid(), lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int i = 0; i < C::TI; ++i)
#pragma unroll
    for (int j = 0; j < C::TJ; ++j) acc[i][j] = (v4d){0., 0., 0., 0.};
  if (nk <= 0) return;
  v2d ra[1][C::CHA], rb[1][C::CHB];
  const unsigned offA = (unsigned)(((t >> 3) * lda + (t & 7) * 2) * (int)sizeof(double));
  const unsigned offB = (unsigned)(((t >> 3) * ldb + (t & 7) * 2) * (int)sizeof(double));
  auto load = [&](auto U_, int kt) {
    constexpr int u = decltype(U_)::value;
#pragma unroll
    for (int q = 0; q < C::CHA; ++q)
      ra[u][q] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Ag + (size_t)q * (C::NT / 8) * lda + (size_t)kt * BK) + offA);
#pragma unroll
    for (int q = 0; q < C::CHB; ++q)
      rb[u][q] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(Bg + (size_t)q * (C::NT / 8) * ldb + (size_t)kt * BK) + offB);
  };
  const int st_off = (t >> 3) * BK + (((t & 7) ^ q_swz((t >> 3) & 15)) << 1);
  auto store = [&](auto U_, double* st) {
    constexpr int u = decltype(U_)::value;
#pragma unroll
    for (int q = 0; q < C::CHA; ++q) *reinterpret_cast<v2d*>(st + st_off + q * (C::NT / 8) * BK) = ra[u][q];
#pragma unroll
    for (int q = 0; q < C::CHB; ++q) *reinterpret_cast<v2d*>(st + BM * BK + st_off + q * (C::NT / 8) * BK) = rb[u][q];
  };
  const int fo0 = fr * BK + (((2 * fk) ^ q_swz(fr)) << 1);
  v2d fa[2][C::TI], fb[2][C::TJ];
  auto frag = [&](auto SET_, const double* st, int h) {
    constexpr int set = decltype(SET_)::value;
    const int fo = fo0 ^ (h << 1);
#pragma unroll
    for (int i = 0; i < C::TI; ++i) fa[set][i] = *reinterpret_cast<const v2d*>(st + (i * WR + wr) * 16 * BK + fo);
#pragma unroll
    for (int j = 0; j < C::TJ; ++j) fb[set][j] = *reinterpret_cast<const v2d*>(st + BM * BK + (wc * C::TJ + j) * 16 * BK + fo);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  const int last = nk - 1;
  load(I0(), 0);
  store(I0(), smem);
  load(I0(), min(1, last));
  store(I0(), smem + STAGE);
  load(I0(), min(2, last));
  __syncthreads();
  frag(I0(), smem, 0);
  int cur = 0;
  // one k-step with the sub-tiles [S, E) of this wave; the ONE staging register set holds step kt + 2 (requested a step ago -- with two sets,
  // two steps ahead as in mainloop_q, the 22 step bodies of the phase dispatch spilled 120 dwords into their loops)
  auto step = [&](int kt, auto S_, auto E_) {
    constexpr int S = decltype(S_)::value, E = decltype(E_)::value;
    const double* st = smem + cur * STAGE;
    const int nxt = cur == 2 ? 0 : cur + 1, wrt = nxt == 2 ? 0 : nxt + 1;
    frag(I1(), st, 1);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = S; i < E; ++i)
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][i][e], fb[0][j][e], acc[i][j], 0, 0, 0);
    store(I0(), smem + wrt * STAGE);
    load(I0(), min(kt + 3, last));
    frag(I0(), smem + nxt * STAGE, 0);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = S; i < E; ++i)
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1][i][e], fb[1][j][e], acc[i][j], 0, 0, 0);
    if constexpr (E > S) {
      constexpr int NM = 2 * (E - S) * C::TJ, NR = C::TI + C::TJ, NW = C::CHA + C::CHB;
      sched_half<0, NM, NR, NW, NW>();
      sched_half<0, NM, NR, 0, 0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    cur = nxt;
  };
  int kt = 0;
  auto run = [&](int end, auto S_, auto E_) {
    for (; kt < end; ++kt) step(kt, S_, E_);
  };
  auto phases = [&](auto E_) {
    constexpr int E = decltype(E_)::value;
    if (E > 0) run(min(nk, nk_full + 0 * WR + wr + 1), std::integral_constant<int, 0>(), E_);
    if (E > 1) run(min(nk, nk_full + 1 * WR + wr + 1), std::integral_constant<int, 1>(), E_);
    if (E > 2) run(min(nk, nk_full + 2 * WR + wr + 1), std::integral_constant<int, 2>(), E_);
    if (E > 3) run(min(nk, nk_full + 3 * WR + wr + 1), std::integral_constant<int, 3>(), E_);
  };
  int n_act = 0;     // sub-tiles of this wave that contain real rows
#pragma unroll
  for (int i = 0; i < C::TI; ++i) n_act += ((i * WR + wr) * 16 < a_rows) ? 1 : 0;
  if (n_act == 4) phases(std::integral_constant<int, 4>());
  else if (n_act == 3) phases(std::integral_constant<int, 3>());
  else if (n_act == 2) phases(std::integral_constant<int, 2>());
  else if (n_act == 1) phases(std::integral_constant<int, 1>());
  run(nk, I0(), I0());               // (what is left only moves the wave's share of the operand tiles)
}

// f(row_in_tile, col_in_tile, value) over the accumulator fragment of mainloop_w
template <int WC, int TI, int TJ, typename F>
__device__ __forceinline__ void for_each_acc_w(v4d (&acc)[TI][TJ], F f) {
  const int tt = mogp_tid(), lane = tt & 63, wave = tt >> 6;
  const int wr = wave / WC, wc = wave % WC;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) f(wr * 16 * TI + i * 16 + (lane >> 4) + 4 * r, wc * 16 * TJ + j * 16 + (lane & 15), acc[i][j][r]);
}

}  // namespace mogp
