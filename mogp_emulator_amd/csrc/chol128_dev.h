// 128 x 128 diagonal block of the blocked Cholesky, factored by ONE 256-thread workgroup entirely on the matrix cores.
// Replaces the diagonal-block share of cusolverDnDpotrf (densegp_gpu.hpp:451-474) / LAPACK dpotrf
// (linalg/cholesky.py:225-232).
//
// The block is an 8 x 8 grid of 16 x 16 sub-blocks.  Its lower triangle lives in REGISTERS for the whole kernel, in
// the fp64 MFMA accumulator layout (lane = (rg, cl) = (lane>>4, lane&15), register q: element [rg + 4q][cl]): wave w
// owns the block rows w and 7-w (nine sub-blocks); off-diagonal sub-blocks are held transposed.  Sub-blocks are loaded
// from global memory straight into that layout (no staging pass, no barrier before the first column step), finished
// columns of L leave for global memory as they are produced; LDS only carries them to the other waves (operands of
// the rank-16 updates) together with the next diagonal sub-block.
//
// One column step j of block step b is a rank-1 update, and a rank-1 update of a SYMMETRIC accumulator needs no data
// movement on this layout: row j of the diagonal sub-block sits in register j>>2 of the 16 lanes with rg == j&3,
// indexed by cl -- exactly where the MFMA expects k-slice j&3 of both its A and its B operand.  So
//     v  = (rg == j&3 && cl >= j) ? D[j][cl] * rsqrt(D[j][j]) : 0        (column j of L, zero in the other k-slices)
//     D  = mfma(-v, v, D)                                                 (D -= v v^T)
// and for a sub-block below it, held transposed (T[c][x] = A[16r + x][16b + c]):
//     u  = (rg == j&3) ? T[j][cl] * rsqrt(D[j][j]) : 0                    (column j of the L sub-block)
//     T  = mfma(-v, u, T)                                                 (T -= v u^T)
// The same sweep applied to an identity block yields inv(L_bb)^T, which the MFMA panel solve below the block wants.
// Every wave factors its own copy of the diagonal sub-block (identical arithmetic, so nothing is exchanged inside a
// block step) and carries two transposed sub-blocks through the 16 steps: three independent MFMAs per step, straight-line
// code.  The panel below the diagonal sub-block is thereby solved by the same backward-stable recurrence as dpotf2 /
// dtrsm.  The pivot of step j+1 does not wait for the MFMA of step j: d' = D[j+1][j+1] - v[j+1]^2 is formed from two
// v_readlane and one FMA, so the dependent chain per column is readlane -> fma -> v_rsq_f64 + 5 FMAs -> multiply
// (~200 cycles; a dependent fp64 MFMA alone costs ~150).  Between block steps: one barrier, the rank-16 update of the
// owned trailing sub-blocks (independent accumulators interleaved), the owner publishes the next diagonal sub-block,
// one barrier.
//
// Round 4: the 16 columns of a block step run as four GROUPS of four (c128_column_groups below; the column-at-a-time form
// described above is kept as c128_column_steps, -DC128_RANK1): all four k-slices of the MFMA carry a column, the 4 x 4 diagonal
// piece of a group is factored on the vector ALU.  Stand-alone kernel 34.9 -> 27.3 us (column steps 18.7 -> 12.0 us of it), the
// diagonal-block task of the one-launch Cholesky 27.7 -> 20.8 us.
//
// Measured on MI355X (tools/chol128_probe.hip): see DESIGN.md section 3 and HISTORY.md.  History: lane = row with pivots and multipliers
// by v_readlane, 2 x 64 dependent column steps and a global-memory round trip between the halves: 58 us; first MFMA
// version (LDS-staged load / store phases at one CU's ~10 B/cycle, MFMA results read back after every instruction): 57 us.
#pragma once
#include "launch.h"

namespace mogp {

typedef double c128_v2d __attribute__((ext_vector_type(2)));
typedef double c128_v4d __attribute__((ext_vector_type(4)));
typedef unsigned int c128_u4 __attribute__((ext_vector_type(4)));

// Write-through (sc1) 16-byte stores for data that ANOTHER workgroup of the same launch reads after a flag hand-off (the
// one-launch Cholesky, kernels_mchol.hip): cdna_hip_programming.md guideline 16, recipe R1 -- payload stored sc1 through a
// buffer descriptor, every storing wave drains (s_waitcnt vmcnt(0)), then one lane stores the flag.  Without SC1 the helper
// is a plain store (separate launches: the kernel boundary publishes).
struct Sc1Buf {
  __amdgpu_buffer_rsrc_t rs;
  const char* base;
};
__device__ __forceinline__ Sc1Buf sc1_buf(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;     // wave-uniform by construction: say so
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  void* ub = (void*)(((unsigned long long)hi << 32) | lo);
  Sc1Buf r;
  r.rs = __builtin_amdgcn_make_buffer_rsrc(ub, 0, bytes, 0x00020000);
  r.base = (const char*)ub;
  return r;
}
template <bool SC1>
__device__ __forceinline__ void st16(const Sc1Buf& b, double* p, c128_v2d x) {
  if (SC1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(c128_u4, x), b.rs, (unsigned)((const char*)p - b.base), 0, 16);
  else *reinterpret_cast<c128_v2d*>(p) = x;
}

__device__ __forceinline__ double readlane_f64(double x, int srclane) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(d) on the critical path of every column: v_rsq_f64 (2^29 ulp, i.e. ~2^-23 relative) and ONE third-order
// correction.  With E = 1 - d r0^2 (|E| ~ 2^-22):  d^-1/2 = r0 (1 - E)^-1/2 = r0 (1 + E/2 + 3E^2/8 + O(E^3)); the
// neglected term is ~2^-67.  Five dependent operations instead of the eight of two coupled Goldschmidt steps; the
// result is within ~1 ulp (the probe compares the factor with a host Cholesky).
__device__ __forceinline__ double rsqrt_fast(double d) {
  const double r0 = __builtin_amdgcn_rsq(d);
  const double t = d * r0;
  const double E = __builtin_fma(-t, r0, 1.0);
  const double p = __builtin_fma(0.375, E, 0.5);
  return __builtin_fma(r0, E * p, r0);
}

// optional cycle stamps for tools/chol128_probe.hip (compiled out of the library)
#ifdef C128_PROFILE
#define C128_STAMP(i) do { if (threadIdx.x == 0 && c128_stamps) c128_stamps[blockIdx.x * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#if C128_PROFILE > 1
#define C128_STAMPW(i) do { if ((threadIdx.x & 63) == 0 && c128_stamps) c128_stamps[2048 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define C128_STAMPW(i) do { } while (0)
#endif
__device__ unsigned long long* c128_stamps = nullptr;
#else
#define C128_STAMP(i) do { } while (0)
#define C128_STAMPW(i) do { } while (0)
#endif

// LDS: the 16 finished columns of the current block step (128 rows, row stride 18 doubles: conflict-free ds_read_b64 in
// both MFMA operand layouts; the two barriers of a block step separate its readers from the next step's writers) + the
// published diagonal sub-block + the inverse staging: 22.5 KB, so that the workgroup fits next to the three resident
// workgroups per CU of a concurrent MFMA update kernel (the first version kept the whole 128 x 130 image, 143 KB, and
// waited for entire CUs to drain: 34 -> 80-105 us under a trailing update).
constexpr int C128_LD = 18;
constexpr int C128_LDS_DOUBLES = 128 * C128_LD + 256 + 256;

// Pack written for the panel solve below the block (trsm128_lds_dev), per emulator:
//   [PACK128_LT  + c * 128 + r]            = L[r][c]                    (transposed, so that lanes run over rows)
//   [PACK128_INV + b * 256 + k * 16 + i]   = inv(L_bb)[i][k]            (the eight 16 x 16 diagonal sub-blocks)
constexpr int PACK128_LT = 0;
constexpr int PACK128_INV = 128 * 128;
constexpr int PACK128_STRIDE = PACK128_INV + 8 * 256;

// 16 column steps of one block step.  All accumulators are held NEGATED (Dn = -D, ...), so that every update is a
// plain accumulate and every sign rides on a free source modifier of a VALU multiply:
//   Dn = this wave's copy of the diagonal sub-block (symmetric), Xn / Yn = two sub-blocks carried along in transposed
//   form (a panel sub-block below the diagonal one, minus the identity for the inverse, or zeros).
// Returns the columns of L (Lv: diagonal sub-block; UX / UY: columns of the solved X / Y) as lane (rg, cl), register
// s = column rg + 4s, row cl.  NSLOT = 1: only Xn is carried.  A pivot that is not > 0 (or NaN / Inf) is recorded in
// `fail` (first one wins) and the factorisation runs on with garbage -- the caller discards the emulator.
template <int NSLOT>
__device__ __forceinline__ void c128_column_steps(c128_v4d& Dn, c128_v4d& Xn, c128_v4d& Yn, c128_v4d& Lv, c128_v4d& UX, c128_v4d& UY, double& rs_last,
                                                  int rg, int cl) {
  // A wave issues in order and has its SIMD to itself, so every VALU instruction counts (~20 per step) and the order
  // written here is the order that matters (pinned for the scheduler by the sched_group_barrier pattern below): the
  // MFMA on the diagonal sub-block first, the reciprocal square root of the NEXT pivot -- which does not depend on it --
  // underneath, then the two carried sub-blocks, and only then the accumulator is read back for the next step.
  //   * one lane mask per step: nrs = (rg == j&3) ? -rs : 0 scales row j of all three accumulators.  Entries cl < j of
  //     the diagonal sub-block's row are already eliminated (rounding residue); they only touch dead rows / columns
  //     and are cleared once per block step by the caller, not per step.
  //   * no pivot test: a pivot <= 0, NaN or Inf turns rs into NaN (v_rsq_f64 of 0 / negative / Inf, then 0 x Inf), and
  //     NaN then reaches every later pivot; the caller tests the last rs of the block.
  double d = -readlane_f64(Dn[0], 0);
  double rs = rsqrt_fast(d);
  double vv = Dn[0] * ((rg == 0) ? -rs : 0.0);
  {
    const double m = -readlane_f64(Dn[0], 1) * rs;
    d = __builtin_fma(-m, m, -readlane_f64(Dn[0], 17));
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int q = j & 3, reg = j >> 2;
    const double nrs = (rg == q) ? -rs : 0.0;              // of step j
    Dn = __builtin_amdgcn_mfma_f64_16x16x4f64(vv, vv, Dn, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (j < 15) rs = rsqrt_fast(d);                        // of step j + 1; independent of the MFMA above
    const double ux = Xn[reg] * nrs;                       // Xn as the MFMA of step j - 1 left it (issued a whole step ago)
    __builtin_amdgcn_sched_barrier(0);
    Xn = __builtin_amdgcn_mfma_f64_16x16x4f64(vv, ux, Xn, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    Lv[reg] += vv;                                         // vv, ux, uy are exact zeros outside the lanes rg == q
    UX[reg] += ux;
    if (NSLOT > 1) {
      const double uy = Yn[reg] * nrs;
      __builtin_amdgcn_sched_barrier(0);
      Yn = __builtin_amdgcn_mfma_f64_16x16x4f64(vv, uy, Yn, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      UY[reg] += uy;
    }
    if (j < 15) {
      // by now the MFMA on Dn has retired (one or two MFMA issue slots later): read it back for step j + 1
      const int jn = j + 1, qn = jn & 3, rn = jn >> 2;
      vv = Dn[rn] * ((rg == qn) ? -rs : 0.0);              // column jn of L: row jn of D / sqrt(pivot)
      if (jn < 15) {
        // pivot after next, D[jn+1][jn+1] - L[jn+1][jn]^2, from two v_readlane and two FMAs: the MFMA of step jn is not waited for
        const double m = -readlane_f64(Dn[rn], 16 * qn + jn + 1) * rs;
        d = __builtin_fma(-m, m, -readlane_f64(Dn[(jn + 1) >> 2], 16 * ((jn + 1) & 3) + jn + 1));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  rs_last = rs;
}

// The same 16 columns in four GROUPS of four (the default; -DC128_RANK1 keeps the column-at-a-time form above): the four k-slices of
// one MFMA carry four columns, so a group costs 2 + 2 NSLOT MFMAs instead of 4 (1 + NSLOT).  Per group g (columns j0 = 4g .. j0+3):
//   * the 4 x 4 diagonal piece G of the group -- register g of the lanes (rg, cl = j0 + b) -- is fetched with v_readlane and factored
//     on the vector ALU with wave-uniform values (G = L44 L44^T: four dependent reciprocal square roots);
//   * lane (rg, cl) forms W[cl][rg], W = inv(L44), by forward substitution on the unit vector e_rg (lanes cl >= 4: zero);
//   * rows j0 .. j0+3 of the symmetric accumulator ARE the MFMA's B operand as they lie (k = rg): one MFMA with A = -W gives
//     L[cl][j0 + rg] = sum_m W[rg][m] D[j0 + m][cl] in output register 0 -- exactly the layout of the rank-4 update's operands --,
//     and the same MFMA on the carried sub-blocks gives their four solved columns;
//   * D += V V^T, X += V UX^T, Y += V UY^T: one MFMA each with all four k-slices used (skipped after the last group).
// The arithmetic differs from the column-at-a-time recurrence by the explicit 4 x 4 inverse (the panel solve below the block uses
// explicit 16 x 16 inverses of the same matrices) and by the MFMA's summation of four products at once.
template <int NSLOT>
__device__ __forceinline__ void c128_column_groups(c128_v4d& Dn, c128_v4d& Xn, c128_v4d& Yn, c128_v4d& Lv, c128_v4d& UX, c128_v4d& UY, double& rs_last,
                                                   int rg, int cl) {
  const c128_v4d zero4 = {0., 0., 0., 0.};
  const double e0 = rg == 0 ? 1.0 : 0.0, e1 = rg == 1 ? 1.0 : 0.0, e2 = rg == 2 ? 1.0 : 0.0, e3 = rg == 3 ? 1.0 : 0.0;
  const double c0 = cl == 0 ? -1.0 : 0.0, c1 = cl == 1 ? -1.0 : 0.0, c2 = cl == 2 ? -1.0 : 0.0, c3 = cl == 3 ? -1.0 : 0.0;
  double rs3 = 1.0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int j0 = 4 * g;
    const double g00 = -readlane_f64(Dn[g], j0), g10 = -readlane_f64(Dn[g], 16 + j0), g11 = -readlane_f64(Dn[g], 16 + j0 + 1);
    const double g20 = -readlane_f64(Dn[g], 32 + j0), g21 = -readlane_f64(Dn[g], 32 + j0 + 1), g22 = -readlane_f64(Dn[g], 32 + j0 + 2);
    const double g30 = -readlane_f64(Dn[g], 48 + j0), g31 = -readlane_f64(Dn[g], 48 + j0 + 1), g32 = -readlane_f64(Dn[g], 48 + j0 + 2),
                 g33 = -readlane_f64(Dn[g], 48 + j0 + 3);
    const double rs0 = rsqrt_fast(g00);
    const double l10 = g10 * rs0, l20 = g20 * rs0, l30 = g30 * rs0;
    const double rs1 = rsqrt_fast(__builtin_fma(-l10, l10, g11));
    const double l21 = __builtin_fma(-l20, l10, g21) * rs1, l31 = __builtin_fma(-l30, l10, g31) * rs1;
    const double rs2 = rsqrt_fast(__builtin_fma(-l21, l21, __builtin_fma(-l20, l20, g22)));
    const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, g32)) * rs2;
    rs3 = rsqrt_fast(__builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, g33))));
    // column rg of W = inv(L44)
    const double w0 = e0 * rs0;
    const double w1 = __builtin_fma(-l10, w0, e1) * rs1;
    const double w2 = __builtin_fma(-l21, w1, __builtin_fma(-l20, w0, e2)) * rs2;
    const double w3 = __builtin_fma(-l32, w2, __builtin_fma(-l31, w1, __builtin_fma(-l30, w0, e3))) * rs3;
    const double nW = __builtin_fma(c3, w3, __builtin_fma(c2, w2, __builtin_fma(c1, w1, c0 * w0)));     // -W[cl][rg]
    __builtin_amdgcn_sched_barrier(0);
    const c128_v4d oD = __builtin_amdgcn_mfma_f64_16x16x4f64(nW, Dn[g], zero4, 0, 0, 0);
    const c128_v4d oX = __builtin_amdgcn_mfma_f64_16x16x4f64(nW, Xn[g], zero4, 0, 0, 0);
    c128_v4d oY = zero4;
    if (NSLOT > 1) oY = __builtin_amdgcn_mfma_f64_16x16x4f64(nW, Yn[g], zero4, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    Lv[g] = oD[0];
    UX[g] = oX[0];
    if (NSLOT > 1) UY[g] = oY[0];
    if (g < 3) {
      Dn = __builtin_amdgcn_mfma_f64_16x16x4f64(oD[0], oD[0], Dn, 0, 0, 0);
      Xn = __builtin_amdgcn_mfma_f64_16x16x4f64(oD[0], oX[0], Xn, 0, 0, 0);
      if (NSLOT > 1) Yn = __builtin_amdgcn_mfma_f64_16x16x4f64(oD[0], oY[0], Yn, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  rs_last = rs3;
}

#ifdef C128_RANK1
#define C128_COLUMNS c128_column_steps
#else
#define C128_COLUMNS c128_column_groups
#endif

// A: origin of the 128 x 128 block (row stride ld); only its lower triangle is read.  On return A holds L (upper
// triangle of the two 64 x 64 diagonal tiles zeroed), pk the pack above, *info_slot = c0 + 1 if it was 0 and the block is not
// positive definite (the failing column inside the block is not recorded: the engine only tests for non-zero).  All 256 threads must call; lds: C128_LDS_DOUBLES doubles.
// SC1_PACK: the pack is stored write-through (see st16) because the panel solves that read it belong to the same launch.
// PRE (one-launch Cholesky): before it is factored, the block receives the rank-128 update with the panel to its LEFT,
// A -= X X^T, X = the 128 x 128 tile at A - 128 (rows of this block, previous block column).  X is streamed through two
// LDS images of 16 columns each and applied to the register-resident sub-blocks with the MFMA pattern of the rank-16
// updates below, so the last panel's share of the diagonal block costs no read-modify-write pass and no hand-off of its
// own.  lds must then hold C128_LDS_PRE_DOUBLES.
// PROG (one-launch Cholesky): *prog = b is published (relaxed agent-scope store by one lane) once the pack entries of the block
// steps < b -- inv(L_b'b') and the columns 16b' .. 16b'+15 of L^T -- are visible to other workgroups, so that the panel solves
// of the dependent chain can run one block step behind the factorisation instead of after it (trsm128_lds_dev<.., PIPE>).  The
// publication costs nothing on the critical path: it happens after the first barrier of block step b, when the only stores a
// wave still has in flight are those of block step b-1 (a whole block step old; the inverse block's stores are issued after that
// barrier for this reason), plus one extra barrier.  The caller publishes 8 after its final drain.
// PRE with a WAIT: the panel to the left is still being solved -- wait(j) returns once its columns 16j .. 16j+15 (all 128 rows)
// are visible (false: the launch has been aborted, and chol128_dev returns false), wait.known(j) says whether that is known
// already without asking.  The update then runs in step with the panel solve that produces X (trsm128_lds_dev<.., PUB>) and is
// complete ~2 us after the solve instead of 14 us after it.  Pieces that are known to be there are requested two steps ahead as
// before; a piece that has to be waited for is requested after the arithmetic of the current one.  after_pre() is called by all
// threads when the update is done (the one-launch Cholesky stamps its trace there).
struct C128NoWait {
  __device__ __forceinline__ bool known(int) const { return true; }
  __device__ __forceinline__ bool operator()(int) { return true; }
};
struct C128Nop {
  __device__ __forceinline__ void operator()() const {}
};
constexpr int C128_LDS_PRE_DOUBLES = C128_LDS_DOUBLES + 128 * C128_LD;
template <bool SC1_PACK = false, bool PRE = false, bool PROG = false, class WAIT = C128NoWait, class AFTER = C128Nop>
__device__ __forceinline__ bool chol128_dev(double* __restrict__ A, int ld, double* __restrict__ pk, int* info_slot, int c0, double* lds,
                                            unsigned* prog = nullptr, WAIT wait = WAIT(), AFTER after_pre = AFTER()) {
  Sc1Buf pkb;
  if (SC1_PACK) pkb = sc1_buf(pk, PACK128_STRIDE * sizeof(double));
  const int t = mogp_tid(), lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int rg = lane >> 4, cl = lane & 15;
  double* Dbuf = lds + 128 * C128_LD;        // next diagonal sub-block (negated), accumulator layout
  double* Ibuf = Dbuf + 256;                 // staging of the inverted diagonal sub-block (one wave per block step)
  const int r1 = w, r2 = 7 - w;
  const c128_v4d zero4 = {0., 0., 0., 0.};
  __builtin_amdgcn_s_setprio(3);             // latency-bound chain: issue ahead of the MFMA waves of a concurrent update kernel
  C128_STAMP(0);
  // sub-block (r, k), k < r, transposed and negated: register q = -A[16r + cl][16k + rg + 4q]
  auto load_off = [&](int r, int k) {
    c128_v4d x;
    const double* p = A + (size_t)(16 * r + cl) * ld + 16 * k + rg;
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = -p[4 * q];
    return x;
  };
  // diagonal sub-block (r, r), both triangles, mirrored from the lower one, negated
  auto load_diag = [&](int r) {
    c128_v4d x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = rg + 4 * q, hi = i > cl ? i : cl, lo = i > cl ? cl : i;
      x[q] = -A[(size_t)(16 * r + hi) * ld + 16 * r + lo];
    }
    return x;
  };
  c128_v4d Dn = load_diag(0);
  c128_v4d R1[4], R2[8];                     // R1[k] = -(sub-block (r1, k)), R2[k] = -(sub-block (r2, k)); block column 0 first
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    R1[k] = zero4;
    if (k < r1) R1[k] = load_off(r1, k);
    else if (k == r1 && k > 0) R1[k] = load_diag(r1);
    R2[k] = load_off(r2, k);                 // r2 >= 4 > k
  }
#pragma unroll
  for (int k = 4; k < 8; ++k) {
    R2[k] = zero4;
    if (k < r2) R2[k] = load_off(r2, k);
    else if (k == r2) R2[k] = load_diag(r2);
  }
  // zeros above the diagonal inside the two 64 x 64 diagonal tiles (three sub-blocks per wave)
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const c128_v2d z2 = {0., 0.};
    if (k > r1 && k <= 3) {
      double* p = A + (size_t)(16 * r1 + cl) * ld + 16 * k + 4 * rg;
      *reinterpret_cast<c128_v2d*>(p) = z2;
      *reinterpret_cast<c128_v2d*>(p + 2) = z2;
    }
    if (k > r2) {
      double* p = A + (size_t)(16 * r2 + cl) * ld + 16 * k + 4 * rg;
      *reinterpret_cast<c128_v2d*>(p) = z2;
      *reinterpret_cast<c128_v2d*>(p + 2) = z2;
    }
  }
  if (PRE) {
    double* img[2] = {lds, lds + C128_LDS_DOUBLES};
    const double* Xg = A - 128;                                 // X[row][col] = Xg[row * ld + col]
    const int xr = t >> 3, xc = 2 * (t & 7);                    // this thread's 16-byte pieces: rows xr + 32 i
    c128_v2d xv[2][4];
    auto xload = [&](int u, int j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[u][i] = *reinterpret_cast<const c128_v2d*>(Xg + (size_t)(xr + 32 * i) * ld + 16 * j + xc);
    };
    if (!wait(0)) return false;
    xload(0, 0);
    if (!wait(1)) return false;
    xload(1, 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double* S = img[j & 1];
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<c128_v2d*>(S + (xr + 32 * i) * C128_LD + xc) = xv[j & 1][i];
      __syncthreads();                                          // (image j & 1 was last read in step j - 2: a barrier ago)
      const bool ahead = j + 2 < 8 && wait.known(j + 2);
      if (ahead) xload(j & 1, j + 2);
      double bo1[4], bo2[4], a0[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bo1[s] = S[(16 * r1 + cl) * C128_LD + rg + 4 * s];
        bo2[s] = S[(16 * r2 + cl) * C128_LD + rg + 4 * s];
        a0[s] = S[cl * C128_LD + rg + 4 * s];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) Dn = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], a0[s], Dn, 0, 0, 0);
      // (round 5: the four operands of sub-block row k + 1 are requested before the MFMAs of row k -- the wave-uniform branches below cut the
      // loop into basic blocks, and inside one the compiler reads, waits, multiplies: the image's latency stood in front of every quad)
      double an[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) an[s] = S[cl * C128_LD + rg + 4 * s];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        double ak[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) ak[s] = an[s];
        if (k + 1 < 8) {
#pragma unroll
          for (int s = 0; s < 4; ++s) an[s] = S[(16 * (k + 1) + cl) * C128_LD + rg + 4 * s];
        }
        __builtin_amdgcn_sched_barrier(0);
        // (sub-blocks the wave does not own, k > r, are skipped with wave-uniform branches: 40 instead of 52 MFMAs per chunk)
        if (k < 4 && k <= r1 && r1 > 0) {
#pragma unroll
          for (int s = 0; s < 4; ++s) R1[k < 4 ? k : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ak[s], bo1[s], R1[k < 4 ? k : 0], 0, 0, 0);
        }
        if (k <= r2) {
#pragma unroll
          for (int s = 0; s < 4; ++s) R2[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ak[s], bo2[s], R2[k], 0, 0, 0);
        }
      }
      if (j + 2 < 8 && !ahead) {
        if (!wait(j + 2)) return false;
        xload(j & 1, j + 2);
      }
    }
    __syncthreads();                                            // the first image is the block steps' column image
    after_pre();
  }
  c128_v4d nident;                           // minus the identity, transposed layout (symmetric)
#pragma unroll
  for (int q = 0; q < 4; ++q) nident[q] = (rg + 4 * q == cl) ? -1.0 : 0.0;
  double rs_last = 1.0;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool p1 = r1 > b, p2 = r2 > b;     // the wave owns a sub-block below diagonal sub-block b in block row r1 / r2
    // the inverse of the diagonal sub-block rides along in a wave that has a free slot
    const bool inv = (b < 3) ? (w == 0) : ((b == 3) ? (w == 3) : (r2 == b));
    double* Sb = lds;                                      // this block step's 16 columns of L: [row][column in block]
    if (r2 >= b) {
      if (b > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Dn[q] = Dbuf[(rg + 4 * q) * 16 + cl];
      }
      c128_v4d Lv = zero4, UX = zero4, UY = zero4;
      C128_STAMPW(4 * b);
      if (b < 4) {
        // two slots: X = sub-block (r1, b) -- or the inverse where r1 <= b --, Y = sub-block (r2, b)
        c128_v4d X = zero4;
        if (b < 3 && p1) X = R1[b < 3 ? b : 0];
        if (inv) X = nident;                 // (inv implies !p1)
        c128_v4d Y = R2[b];                  // r2 >= 4 > b
        C128_COLUMNS<2>(Dn, X, Y, Lv, UX, UY, rs_last, rg, cl);
      } else {
        // one slot: sub-block (r2, b), or the inverse in the wave that owns the diagonal sub-block
        c128_v4d X = p2 ? R2[b] : nident, Y = zero4;
        C128_COLUMNS<1>(Dn, X, Y, Lv, UX, UY, rs_last, rg, cl);
        UY = UX;                             // uniform naming below: UY belongs to block row r2
      }
      C128_STAMPW(4 * b + 1);
      // finished columns into the LDS image: lane (rg, cl), register s = L[16r + cl][16b + rg + 4s]
      auto put = [&](int r, const c128_v4d& U) {
        double* ps = Sb + (16 * r + cl) * C128_LD + rg;
#pragma unroll
        for (int s = 0; s < 4; ++s) ps[4 * s] = U[s];
      };
      if (r1 == b || r2 == b) {
        // the strict upper triangle of the diagonal sub-block as exact zeros (see c128_column_steps)
#pragma unroll
        for (int q = 0; q < 4; ++q) Lv[q] = (cl >= rg + 4 * q) ? Lv[q] : 0.0;
        put(b, Lv);
      }
      if (b < 3 && p1) put(r1, UX);
      if (p2) put(r2, UY);
      if (inv) {
        // columns of inv(L_bb)^T: lane (rg, cl), register s = inv(L_bb)[rg + 4s][cl]; pack order [k = cl][i = rg + 4s];
        // through a wave-private LDS stage so that it leaves as two fully coalesced 1 KB stores
        const c128_v4d& UI = (b < 4) ? UX : UY;
        double* stage = Ibuf;
#pragma unroll
        for (int s = 0; s < 4; ++s) stage[cl * 16 + rg + 4 * s] = UI[s];
        __builtin_amdgcn_wave_barrier();
        if (!PROG) {
          double* pi = pk + PACK128_INV + b * 256;
#pragma unroll
          for (int h = 0; h < 2; ++h)
            st16<SC1_PACK>(pkb, pi + 128 * h + 2 * lane, *reinterpret_cast<const c128_v2d*>(stage + 128 * h + 2 * lane));
        }
      }
    }
    C128_STAMPW(4 * b + 2);
    __syncthreads();
    C128_STAMPW(4 * b + 3);
    C128_STAMP(2 + 2 * b);
    if (PROG) {
      if (b > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (stores of block step b - 1: long landed)
        __syncthreads();
        if (t == 0) __hip_atomic_store(prog, (unsigned)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (inv) {                                                  // the inverse block staged above (same wave), now stored
        double* pi = pk + PACK128_INV + b * 256;
#pragma unroll
        for (int h = 0; h < 2; ++h)
          st16<SC1_PACK>(pkb, pi + 128 * h + 2 * lane, *reinterpret_cast<const c128_v2d*>(Ibuf + 128 * h + 2 * lane));
      }
    }
    // block column b of L leaves the LDS image as full 128-byte row segments: A (row-major) and the transposed pack.
    // All LDS reads first, then the stores (fire and forget: they drain underneath the MFMAs below).
    {
      const int nrows = 128 - 16 * b;
      c128_v2d ca[4], ct[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (32 * i < nrows) {
          const int row = 16 * b + (t >> 3) + 32 * i, rc = row < 128 ? row : 127;
          ca[i] = *reinterpret_cast<const c128_v2d*>(Sb + rc * C128_LD + 2 * (t & 7));
          const int r = 16 * b + 2 * (t & 15) + 32 * i, rr = r < 128 ? r : 126;
          ct[i][0] = Sb[rr * C128_LD + (t >> 4)];
          ct[i][1] = Sb[(rr + 1) * C128_LD + (t >> 4)];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (32 * i < nrows) {
          const int row = 16 * b + (t >> 3) + 32 * i;
          if (row < 128) *reinterpret_cast<c128_v2d*>(A + (size_t)row * ld + 16 * b + 2 * (t & 7)) = ca[i];
          const int r = 16 * b + 2 * (t & 15) + 32 * i, c = 16 * b + (t >> 4);
          if (r < 128) st16<SC1_PACK>(pkb, pk + PACK128_LT + c * 128 + r, ct[i]);
        }
      }
    }
    if (b == 7) break;
    C128_STAMPW(32 + 4 * b);
    // rank-16 update of the owned sub-blocks to the right of block column b.  All operands are requested from the LDS image
    // first; sub-blocks the wave does not own (k > r) are skipped with wave-uniform branches (wave 0 -- block rows 0 and 7 -- is
    // the longest in every block step: 7 - b sub-blocks instead of 10 - b issued); with k == r the A operand is the wave's own
    // panel sub-block, i.e. the symmetric update of the diagonal sub-block.
    {
      double bo1[4], bo2[4];
      const double* pb1 = Sb + (16 * r1 + cl) * C128_LD + rg;
      const double* pb2 = Sb + (16 * r2 + cl) * C128_LD + rg;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bo1[s] = pb1[4 * s];
        bo2[s] = pb2[4 * s];
      }
      double aa[7][4];
#pragma unroll
      for (int k = b + 1; k < 8; ++k) {
        const double* pa = Sb + (16 * k + cl) * C128_LD + rg;
#pragma unroll
        for (int s = 0; s < 4; ++s) aa[k - b - 1][s] = pa[4 * s];
      }
#pragma unroll
      for (int k = b + 1; k < 8; ++k) {
        if (k < 4 && k <= r1) {
#pragma unroll
          for (int s = 0; s < 4; ++s) R1[k < 4 ? k : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[k - b - 1][s], bo1[s], R1[k < 4 ? k : 0], 0, 0, 0);
        }
        if (k <= r2) {
#pragma unroll
          for (int s = 0; s < 4; ++s) R2[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[k - b - 1][s], bo2[s], R2[k], 0, 0, 0);
        }
      }
    }
    C128_STAMPW(33 + 4 * b);
    // the owner of the next diagonal sub-block publishes it
    {
      const bool own = (b + 1 < 4) ? (r1 == b + 1) : (r2 == b + 1);
      const c128_v4d Dnext = (b + 1 < 4) ? R1[b + 1 < 4 ? b + 1 : 0] : R2[b + 1];
      if (own) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Dbuf[(rg + 4 * q) * 16 + cl] = Dnext[q];
      }
    }
    __syncthreads();
    C128_STAMPW(34 + 4 * b);
    C128_STAMP(3 + 2 * b);
  }
  // wave 0 took part in every block step: its last reciprocal square root is NaN iff some pivot was not a positive finite number
  if (w == 0 && lane == 0 && !(rs_last > 0.0) && *info_slot == 0) *info_slot = c0 + 1;
  C128_STAMP(18);
  return true;
}

}  // namespace mogp
