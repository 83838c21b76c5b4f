// 128-wide panel solve below a factored 128 x 128 diagonal block, on the matrix cores.
#pragma once
#include <type_traits>
#include "launch.h"
#include "chol128_dev.h"

namespace mogp {

typedef double v4d_t __attribute__((ext_vector_type(4)));
typedef double v2d_p __attribute__((ext_vector_type(2)));

// X L^T = B for the rows below the block.  Each wave owns a 16-row slab B (16 x 128) and solves the transposed form
// X^T = L^-1 B^T by block forward substitution over the eight 16-column blocks:
//     T_b = B_b^T - sum_{a<b} L_ba X_a^T ,   X_b^T = inv(L_bb) T_b        (v_mfma_f64_16x16x4)
// The transposed form chains without any LDS transpose: an MFMA result (lane holds rows g+4r, g = lane>>4, column
// lane&15) is used directly as the B operand of the next MFMA with the k index running over g+4r, and the A operand
// (L_ba or inv(L_bb), element [lane&15][g+4r]) has the same k mapping in the pack written by chol128_dev.
// The rank-64 update of the second half of the block column with the first is part of the substitution, so the panel
// is read and written once.
//
// Operands of the diagonal block out of LDS.  The first version (round 1 - first session of round 2) kept the whole slab in
// registers through a 16.6 KB wave-private stage and fetched every MFMA's A operand (512 bytes of the pack) through the
// vector L1: 144 loads = 74 KB per 16-row slab, and with eight waves per CU that is the L1's whole 64 B/clk -- the kernel
// was bound by operand traffic (3.1 TB/s of panel bytes, 601 us per fit of 64 x n=2000), not by HBM or the matrix pipe.
// Here the four waves of a workgroup walk the eight 16-column blocks in step and share ONE image of what step b needs --
// row block b of L (16 x 16b entries, as [column][row]) and inv(L_bb) -- double-buffered and requested a step ahead; the
// slab is not held in registers either: block b of it is fetched when step b needs it (two 16-byte loads per lane
// through a 2.3 KB wave-private transposing stage) and X_b leaves the same way.  51 KB of LDS and 112 VGPRs per
// workgroup instead of 66.5 KB / 176: three workgroups per CU instead of two; 4.6 TB/s in the large launches, 531 us per
// fit, bit-identical results.
constexpr int TL_PK = 112 * 16 + 256;                 // row block image: A operands of the blocks a < b, then inv(L_bb)
constexpr int TL_TS = 16 * 18;                        // one 16 x 16 block of the slab, row stride 18
constexpr int TRSM128L_LDS = 2 * TL_PK + 4 * 2 * TL_TS;

// SC1_OUT: the solved rows are stored write-through (chol128_dev.h, st16): other workgroups of the same launch read them
// as GEMM operands after a flag hand-off (kernels_mchol.hip).  The emulator's matrix must then be smaller than 4 GB.
// DEEP (one-launch Cholesky, where a panel solve runs next to GEMM tasks and every load sees a loaded memory system): the
// whole 16 x 128 slab is requested up front (32 VGPRs) and the row-block images two steps ahead instead of one -- the
// early steps have MFMA chains of 4 - 12 instructions and were bound by one memory latency each (16 - 21 us per solve
// under load against 8 us alone, tools/mchol_trace.py).  Arithmetic and results are unchanged.
// PIPE (implies DEEP's slab preload): the pack is still being written -- the diagonal block publishes its progress per block step
// (chol128_dev<.., PROG>) and wait(b) returns once row block b of the pack (the L^T columns of the block steps < b and inv(L_bb))
// is visible, or false when the launch has been aborted.  Step b is computed first, then the wait for step b + 1, its request and
// deposit: the solve runs one block step behind the factorisation and ends ~4 us after it instead of ~9.  Returns false on abort.
// PUB (with PIPE): pub(b) is called by all threads once the 16 columns 16b .. 16b+15 of the workgroup's 64 solved rows are
// visible to other workgroups (every wave has drained its write-through stores, then a barrier), b = 0 .. 6; the caller
// publishes the last block with its own final drain.  The diagonal block of the NEXT block column consumes the rows in
// these pieces (chol128_dev<.., PRE> with a wait), two block steps behind the factorisation that feeds this solve.  The
// drain sits in the slack of the solve: it is one block step behind a factorisation that takes ~4 us per step.
struct TrsmNoWait {
  __device__ __forceinline__ bool operator()(int) const { return true; }
};
struct TrsmNoPub {
  __device__ __forceinline__ void operator()(int) const {}
};
template <bool SC1_OUT = false, bool DEEP = false, bool PIPE = false, class WAIT = TrsmNoWait, class PUB = TrsmNoPub>
__device__ __forceinline__ bool trsm128_lds_dev(const BatchView& v, int c0, int r0, const double* __restrict__ pk, int emu, int rowblock,
                                                double* lds, WAIT wait = WAIT(), PUB pub = PUB()) {
  // (PUB without PIPE -- with DEEP: the pack is complete, the images are requested two steps ahead, and the solved rows are still
  // published piece by piece for the next diagonal block: a chain task whose GEMM ended after its diagonal block had finished)
  constexpr bool PUBLISH = (PIPE || DEEP) && !std::is_same<PUB, TrsmNoPub>::value;
  Sc1Buf ab;
  if (SC1_OUT) ab = sc1_buf(v.A + (size_t)emu * v.MS, (unsigned)(v.MS * sizeof(double)));
  const int ld = v.LD;
  const int t = mogp_tid(), lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, i = lane & 15;
  double* slab = v.A + (size_t)emu * v.MS + (size_t)(r0 + rowblock * 64 + wave * 16) * ld + c0;   // 16 rows x 128 columns
  double* pkb[2] = {lds, lds + TL_PK};
  double* tsb[2] = {lds + 2 * TL_PK + wave * 2 * TL_TS, lds + 2 * TL_PK + wave * 2 * TL_TS + TL_TS};
  const double* LT = pk + PACK128_LT;
  // this lane's two 16-byte pieces of a slab block: rows q >> 3, piece q & 7 for q = lane, lane + 64
  const int sr0 = lane >> 3, sp = (lane & 7) * 2;
  constexpr bool PRE_SLAB = DEEP || PIPE;
  v2d_p sl[PRE_SLAB ? 8 : 1][2];
  auto slab_request = [&](int b) {
    sl[PRE_SLAB ? b : 0][0] = *reinterpret_cast<const v2d_p*>(slab + (size_t)sr0 * ld + 16 * b + sp);
    sl[PRE_SLAB ? b : 0][1] = *reinterpret_cast<const v2d_p*>(slab + (size_t)(sr0 + 8) * ld + 16 * b + sp);
  };
  auto slab_deposit = [&](int b, double* ts) {
    *reinterpret_cast<v2d_p*>(ts + sr0 * 18 + sp) = sl[PRE_SLAB ? b : 0][0];
    *reinterpret_cast<v2d_p*>(ts + (sr0 + 8) * 18 + sp) = sl[PRE_SLAB ? b : 0][1];
  };
  // the workgroup's share of row block b: 8 b pieces of 16 doubles [column c][rows 16b .. 16b+15], then the 256 doubles of inv(L_bb)
  v2d_p pr[DEEP ? 2 : 1][4], pinv[DEEP ? 2 : 1];
  auto pack_request = [&](int b) {
    const int u = DEEP ? (b & 1) : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * b) pr[u][q] = *reinterpret_cast<const v2d_p*>(LT + (size_t)(ch >> 3) * 128 + 16 * b + (ch & 7) * 2);
    }
    if (t < 128) pinv[u] = *reinterpret_cast<const v2d_p*>(pk + PACK128_INV + b * 256 + 2 * t);
  };
  auto pack_deposit = [&](int b, double* img) {
    const int u = DEEP ? (b & 1) : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * b) *reinterpret_cast<v2d_p*>(img + (ch >> 3) * 16 + (ch & 7) * 2) = pr[u][q];
    }
    if (t < 128) *reinterpret_cast<v2d_p*>(img + 112 * 16 + 2 * t) = pinv[u];
  };
  if (PIPE) {
#pragma unroll
    for (int b = 0; b < 8; ++b) slab_request(b);
    if (!wait(0)) return false;
    pack_request(0);
  } else {
    pack_request(0);
    if (DEEP) {
      pack_request(1);
#pragma unroll
      for (int b = 0; b < 8; ++b) slab_request(b);
    } else {
      slab_request(0);
    }
  }
  pack_deposit(0, pkb[0]);
  slab_deposit(0, tsb[0]);
  __syncthreads();
  v4d_t X[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const double* img = pkb[b & 1];
    double* ts = tsb[b & 1];
    if (PIPE) {
      // (request and deposit of step b + 1 follow the compute of step b, behind the wait)
    } else if (DEEP) {
      if (b + 2 < 8) pack_request(b + 2);        // register set b & 1 held row block b, which is in LDS already
    } else if (b < 7) {
      pack_request(b + 1);
      slab_request(b + 1);
    }
    v4d_t T;
#pragma unroll
    for (int r = 0; r < 4; ++r) T[r] = ts[i * 18 + g + 4 * r];
#pragma unroll
    for (int a = 0; a < b; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r)      // A operand: L[16b + i][16a + g + 4r]
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(-img[(16 * a + g + 4 * r) * 16 + i], X[a][r], T, 0, 0, 0);
    const double* inv = img + 112 * 16;
    X[b] = (v4d_t){0., 0., 0., 0.};
#pragma unroll
    for (int r = 0; r < 4; ++r) X[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(inv[(g + 4 * r) * 16 + i], T[r], X[b], 0, 0, 0);
    // X_b leaves through the stage it came in by: full 128-byte row segments
#pragma unroll
    for (int r = 0; r < 4; ++r) ts[i * 18 + g + 4 * r] = X[b][r];
    __builtin_amdgcn_wave_barrier();
    {
      const v2d_p o0 = *reinterpret_cast<const v2d_p*>(ts + sr0 * 18 + sp);
      const v2d_p o1 = *reinterpret_cast<const v2d_p*>(ts + (sr0 + 8) * 18 + sp);
      st16<SC1_OUT>(ab, slab + (size_t)sr0 * ld + 16 * b + sp, o0);
      st16<SC1_OUT>(ab, slab + (size_t)(sr0 + 8) * ld + 16 * b + sp, o1);
    }
    if (PUBLISH && b < 7) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      pub(b);
    }
    if (b < 7) {
      if (PIPE) {
        if (!wait(b + 1)) return false;
        pack_request(b + 1);
      }
      pack_deposit(b + 1, pkb[(b + 1) & 1]);
      slab_deposit(b + 1, tsb[(b + 1) & 1]);
    }
    __syncthreads();
  }
  return true;
}

// The same solve for a 64 x 128 tile that a GEMM task of the one-launch Cholesky still holds in REGISTERS (x = C - acc in the fp64 MFMA
// accumulator layout of a 2 x 2-wave 64 x 128 tile: wave (wr, wc) rows 32 wr + 16 i + (lane >> 4) + 4 q, columns 64 wc + 16 j + (lane & 15)):
// the tile goes to the solving waves through LDS instead of through global memory (store, drain, barrier, 16-byte re-reads: ~5 us of
// write-back plus the slab latency in front of the first block step, per-task stamps of tools/mchol_trace.py).  Wave w solves the slab
// of rows 16 w .. 16 w + 15 as above; the stage holds HALF a tile -- [4 slabs][4 blocks][16 x 18] -- so the column half 0 .. 63 (held by
// the waves wc = 0) is deposited first, block steps 0 - 3 run, then the waves wc = 1 deposit the other half into the same slots.  Same
// arithmetic in the same order as trsm128_lds_dev: bit-identical results.  lds: TRSM128T_LDS doubles.
constexpr int TRSM128T_LDS = 2 * TL_PK + 16 * TL_TS;
// The images of block steps 0 and 1 of a FINISHED diagonal block's pack, requested by a bulk task BEFORE it reads its C tile (their
// latency then runs under the tile's loads and the subtraction); trsm128_tile2_dev deposits them.
struct TrsmSlabPre {
  v2d_p pr[2][4], pinv[2];
};
__device__ __forceinline__ void trsm128_slab_request(const double* __restrict__ pk, TrsmSlabPre& P) {
  const int t = mogp_tid();
  const double* LT = pk + PACK128_LT;
  // the images of block steps 0 and 1 (trsm128_tile2_dev: column block b of L below its diagonal sub-block, and inv(L_bb))
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * (7 - b)) P.pr[b][q] = *reinterpret_cast<const v2d_p*>(LT + (size_t)(16 * b + ((ch >> 3) & 15)) * 128 + 16 * (b + 1 + (ch >> 7)) + (ch & 7) * 2);
    }
    if (t < 128) P.pinv[b] = *reinterpret_cast<const v2d_p*>(pk + PACK128_INV + b * 256 + 2 * t);
  }
}

// Round 4 (default for bulk tasks): the tile is RE-DEALT to the solving waves before the solve.  All four waves put their 32 x 64 piece of
// x = C - acc into a whole-tile stage at once ([slab][block][16 x 17]: 8704 doubles, exactly the LDS of the round-3 form), one barrier, and
// wave w takes the eight blocks of ITS slab as transposed fragments T[b] into the registers x used to occupy; the pack images move in
// afterwards.  From then on nothing but the images is shared.  Round 3 (trsm128_tile_dev, MOGP_MC_SLAB=0) staged half a tile and
// deposited the column half 64 .. 127 after block step 3: its 32 values per lane had to survive four block steps in the registers of two
// waves, the compiler spilled 18 of them to scratch, and the mid-solve deposit -- reload from scratch memory under load, two barriers --
// took 5.5 - 6 us of a 30 us solve next to a GEMM partner (per-step stamps); the first deposit of a block waited for three other waves
// (10 us before the first block step).  Tried on the way: GEMM on 4 x 1 waves, so that every wave already holds its slab (no re-deal at
// all): solve 30 -> 20 us, but 9 instead of 6 LDS fragment reads per 8 MFMAs made the GEMM phase 4 % slower (n=16000: +3 % overall);
// the whole pack requested before the C tile (24 loads per thread): no gain, more spills; the operands of a block step all requested
// in front of its first MFMA with the sign flips folded into the stage: slower (the chain is not LDS-latency-bound: 160 cycles per MFMA
// next to a GEMM partner whichever way the operands arrive -- the partner's MFMAs take their share of the SIMD's pipe).
// Same arithmetic in the same order as trsm128_lds_dev: bit-identical.  x: C - acc in the accumulator layout of trsm128_tile_dev;
// P: trsm128_slab_request.  lds: TRSM128T_LDS doubles.
constexpr int TL_TS17 = 16 * 17;
static_assert(32 * TL_TS17 <= TRSM128T_LDS, "whole-tile stage must fit the LDS of the half-tile form");
template <bool SC1_OUT>
__device__ __forceinline__ void trsm128_tile2_dev(const BatchView& v, int c0, int r0, const double* __restrict__ pk, int emu, double* lds,
                                                  const v4d_t (&x)[2][4], TrsmSlabPre& P, unsigned long long* stamps = nullptr) {
  Sc1Buf ab;
  if (SC1_OUT) ab = sc1_buf(v.A + (size_t)emu * v.MS, (unsigned)(v.MS * sizeof(double)));
  const int ld = v.LD;
  const int t = mogp_tid(), lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, i = lane & 15;
  const int wr = wave >> 1, wc = wave & 1;
  double* slab = v.A + (size_t)emu * v.MS + (size_t)(r0 + wave * 16) * ld + c0;   // 16 rows x 128 columns
  double* pkb[2] = {lds, lds + TL_PK};
  double* ts = lds + 2 * TL_PK + wave * TL_TS;           // (after the re-deal: this wave's private 16 x 18 stage for the way out)
  const double* LT = pk + PACK128_LT;
  const int sr0 = lane >> 3, sp = (lane & 7) * 2;
  // image of block step b: [0, 256) inv(L_bb) as in the pack; [(16 c + k) * 16 + i], c = b+1 .. 7: L[16 c + i][16 b + k] -- column block b
  // of L below its diagonal sub-block (rows 16 b .. 16 b + 15 of the pack's transposed copy), the A operands of the updates of step b
  auto pack_request = [&](int b) {
    const int u = b & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * (7 - b))
        P.pr[u][q] = *reinterpret_cast<const v2d_p*>(LT + (size_t)(16 * b + ((ch >> 3) & 15)) * 128 + 16 * (b + 1 + (ch >> 7)) + (ch & 7) * 2);
    }
    if (t < 128) P.pinv[u] = *reinterpret_cast<const v2d_p*>(pk + PACK128_INV + b * 256 + 2 * t);
  };
  auto pack_deposit = [&](int b, double* img) {
    const int u = b & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * (7 - b)) *reinterpret_cast<v2d_p*>(img + ((b + 1 + (ch >> 7)) * 16 + ((ch >> 3) & 15)) * 16 + (ch & 7) * 2) = P.pr[u][q];
    }
    if (t < 128) *reinterpret_cast<v2d_p*>(img + 2 * t) = P.pinv[u];
  };
  // whole-tile stage over ALL of lds: [slab 2 wr + ii][block 4 wc + j][16 x 17]
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) lds[((2 * wr + ii) * 8 + 4 * wc + j) * TL_TS17 + (g + 4 * q) * 17 + i] = x[ii][j][q];
  __syncthreads();
  v4d_t T[8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) T[b][r] = lds[(wave * 8 + b) * TL_TS17 + i * 17 + g + 4 * r];
  __syncthreads();               // every wave has taken its slab: the images move in
  pack_deposit(0, pkb[0]);
  __syncthreads();
  if (stamps && t == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();
  v4d_t X[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const double* img = pkb[b & 1];
    if (b + 2 < 8) pack_request(b + 2);
    // RIGHT-LOOKING (round 5): x_b = inv(L_bb) t_b, then every later block takes its update t_c -= L_cb x_b at once.  The left-looking order of
    // rounds 3 - 4 (t_b collects the updates of all earlier blocks in step b) made the solve ONE chain of 144 dependent MFMAs; here a step's
    // chain is its 4 + 4 (x_b, then t_{b+1}) and the other 4 (6 - b) MFMAs are independent of it.  Every t_c still receives its updates in the
    // order b = 0, 1, .. with the same operands: bit-identical to the left-looking forms (trsm128_lds_dev, the chain tasks).
    X[b] = (v4d_t){0., 0., 0., 0.};
#pragma unroll
    for (int r = 0; r < 4; ++r) X[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(img[(g + 4 * r) * 16 + i], T[b][r], X[b], 0, 0, 0);
#pragma unroll
    for (int c = b + 1; c < 8; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-img[(16 * c + g + 4 * r) * 16 + i], X[b][r], T[c], 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) ts[i * 18 + g + 4 * r] = X[b][r];
    __builtin_amdgcn_wave_barrier();
    if (stamps && t == 0) stamps[4 + 2 * b] = __builtin_amdgcn_s_memrealtime();
    {
      const v2d_p o0 = *reinterpret_cast<const v2d_p*>(ts + sr0 * 18 + sp);
      const v2d_p o1 = *reinterpret_cast<const v2d_p*>(ts + (sr0 + 8) * 18 + sp);
      st16<SC1_OUT>(ab, slab + (size_t)sr0 * ld + 16 * b + sp, o0);
      st16<SC1_OUT>(ab, slab + (size_t)(sr0 + 8) * ld + 16 * b + sp, o1);
    }
    __builtin_amdgcn_wave_barrier();
    if (b == 3 && stamps && t == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
    if (b < 7) pack_deposit(b + 1, pkb[(b + 1) & 1]);
    __syncthreads();
    if (stamps && t == 0) stamps[5 + 2 * b] = __builtin_amdgcn_s_memrealtime();
  }
  if (stamps && t == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
}

// The re-dealt solve for a CHAIN task (round 4): x never goes to global memory -- it is re-dealt as in trsm128_tile2_dev -- and the block
// steps follow the diagonal block that is still being factored (LATE = false: wait(b) as in trsm128_lds_dev<PIPE>, image b requested behind
// its wait) or, when that block has already finished (LATE = true), run with the images requested two steps ahead.  pub(b) after the 16
// columns of block step b are visible (b = 0 .. 6), as trsm128_lds_dev<PUB>.  Before: x written back (32 stores per lane, drain, barrier)
// and re-read by the solve as 16-byte pieces: ~5 us of the row-band path that is one of the two critical paths of a single-matrix
// factorisation (HISTORY.md).  Same arithmetic in the same order: bit-identical.  Returns false when the launch has been aborted.
template <bool SC1_OUT, bool LATE, class WAIT, class PUB>
__device__ __forceinline__ bool trsm128_tile2_chain_dev(const BatchView& v, int c0, int r0, const double* __restrict__ pk, int emu, double* lds,
                                                        const v4d_t (&x)[2][4], WAIT wait, PUB pub) {
  Sc1Buf ab;
  if (SC1_OUT) ab = sc1_buf(v.A + (size_t)emu * v.MS, (unsigned)(v.MS * sizeof(double)));
  const int ld = v.LD;
  const int t = mogp_tid(), lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, i = lane & 15;
  const int wr = wave >> 1, wc = wave & 1;
  double* slab = v.A + (size_t)emu * v.MS + (size_t)(r0 + wave * 16) * ld + c0;   // 16 rows x 128 columns
  double* pkb[2] = {lds, lds + TL_PK};
  double* ts = lds + 2 * TL_PK + wave * TL_TS;
  const double* LT = pk + PACK128_LT;
  const int sr0 = lane >> 3, sp = (lane & 7) * 2;
  v2d_p pr[2][4], pinv[2];
  auto pack_request = [&](int b) {
    const int u = b & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * b) pr[u][q] = *reinterpret_cast<const v2d_p*>(LT + (size_t)(ch >> 3) * 128 + 16 * b + (ch & 7) * 2);
    }
    if (t < 128) pinv[u] = *reinterpret_cast<const v2d_p*>(pk + PACK128_INV + b * 256 + 2 * t);
  };
  auto pack_deposit = [&](int b, double* img) {
    const int u = b & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * b) *reinterpret_cast<v2d_p*>(img + (ch >> 3) * 16 + (ch & 7) * 2) = pr[u][q];
    }
    if (t < 128) *reinterpret_cast<v2d_p*>(img + 112 * 16 + 2 * t) = pinv[u];
  };
  auto rows_request = [&](int b) {
    const int u = b & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * b) pr[u][q] = *reinterpret_cast<const v2d_p*>(LT + (size_t)(ch >> 3) * 128 + 16 * b + (ch & 7) * 2);
    }
  };
  auto rows_deposit = [&](int b, double* img) {
    const int u = b & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * b) *reinterpret_cast<v2d_p*>(img + (ch >> 3) * 16 + (ch & 7) * 2) = pr[u][q];
    }
  };
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) lds[((2 * wr + ii) * 8 + 4 * wc + j) * TL_TS17 + (g + 4 * q) * 17 + i] = x[ii][j][q];
  __syncthreads();
  v4d_t T[8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) T[b][r] = lds[(wave * 8 + b) * TL_TS17 + i * 17 + g + 4 * r];
  if (LATE) {
    pack_request(0);
    pack_request(1);
    __syncthreads();
    pack_deposit(0, pkb[0]);
  }
  __syncthreads();                       // (every wave has taken its slab)
  v4d_t X[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const double* img = pkb[b & 1];
    if (LATE && b + 2 < 8) pack_request(b + 2);
    // Pipelined form: the updates of block step b only need row block b of L, i.e. the pieces < b: they run BEFORE piece b is
    // waited for, on an image requested during block step b - 1 (behind that step's wait); only inv(L_bb) -- fetched straight
    // into the MFMA operand layout, no LDS image, no barrier -- is read behind the wait for piece b.  (Before: the whole image
    // of step b was requested behind the wait: per block step one exposed round trip plus the deposit plus up to 28 MFMAs.)
    // (round 5: every MFMA of the step is chained on Tb and takes its A operand from an LDS read of its own; left alone the compiler issues
    // each read in front of its MFMA and waits for it there -- ~100 cycles of LDS latency on top of the MFMA's 64, and a chain task is alone
    // on its CU's pipes in the launches where it matters.  The four operands of the NEXT quad are now requested before the MFMAs of this one,
    // pinned by scheduling barriers; X holds -x from here on (the sign on the four values of x_a, not on every operand): same products.)
    v4d_t Tb = T[b];
    if (b > 0) {
      double an[4], ac[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) an[r] = img[(g + 4 * r) * 16 + i];
#pragma unroll
      for (int a = 0; a < b; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ac[r] = an[r];
        if (a + 1 < b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) an[r] = img[(16 * (a + 1) + g + 4 * r) * 16 + i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Tb = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[r], X[a][r], Tb, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    double invr[4];
    if (LATE) {
      const double* inv = img + 112 * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) invr[r] = inv[(g + 4 * r) * 16 + i];
    } else {
      if (!wait(b)) return false;        // (contains barriers)
#pragma unroll
      for (int r = 0; r < 4; ++r) invr[r] = pk[PACK128_INV + b * 256 + (g + 4 * r) * 16 + i];
      if (b + 1 < 8) rows_request(b + 1);      // (row block b + 1 needs the pieces <= b)
    }
    X[b] = (v4d_t){0., 0., 0., 0.};
#pragma unroll
    for (int r = 0; r < 4; ++r) X[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(invr[r], Tb[r], X[b], 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) ts[i * 18 + g + 4 * r] = X[b][r];
    X[b] = -X[b];                          // (only an operand of the later steps from here on)
    __builtin_amdgcn_wave_barrier();
    {
      const v2d_p o0 = *reinterpret_cast<const v2d_p*>(ts + sr0 * 18 + sp);
      const v2d_p o1 = *reinterpret_cast<const v2d_p*>(ts + (sr0 + 8) * 18 + sp);
      st16<SC1_OUT>(ab, slab + (size_t)sr0 * ld + 16 * b + sp, o0);
      st16<SC1_OUT>(ab, slab + (size_t)(sr0 + 8) * ld + 16 * b + sp, o1);
    }
    if (b < 7) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      pub(b);
      if (LATE) pack_deposit(b + 1, pkb[(b + 1) & 1]);
      else rows_deposit(b + 1, pkb[(b + 1) & 1]);
    }
    __syncthreads();
  }
  return true;
}

// stamps (analysis only, MOGP_MC_TRACE): [0] after the first barrier, [1] after block step 3, [2] after block step 7
template <bool SC1_OUT>
__device__ __forceinline__ void trsm128_tile_dev(const BatchView& v, int c0, int r0, const double* __restrict__ pk, int emu, double* lds,
                                                 const v4d_t (&x)[2][4], unsigned long long* stamps = nullptr) {
  Sc1Buf ab;
  if (SC1_OUT) ab = sc1_buf(v.A + (size_t)emu * v.MS, (unsigned)(v.MS * sizeof(double)));
  const int ld = v.LD;
  const int t = mogp_tid(), lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, i = lane & 15;
  const int wr = wave >> 1, wc = wave & 1;
  double* slab = v.A + (size_t)emu * v.MS + (size_t)(r0 + wave * 16) * ld + c0;   // 16 rows x 128 columns
  double* pkb[2] = {lds, lds + TL_PK};
  double* stage = lds + 2 * TL_PK;                       // [slab w][block b & 3][16 x 18]
  const double* LT = pk + PACK128_LT;
  const int sr0 = lane >> 3, sp = (lane & 7) * 2;
  v2d_p pr[2][4], pinv[2];
  auto pack_request = [&](int b) {
    const int u = b & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * b) pr[u][q] = *reinterpret_cast<const v2d_p*>(LT + (size_t)(ch >> 3) * 128 + 16 * b + (ch & 7) * 2);
    }
    if (t < 128) pinv[u] = *reinterpret_cast<const v2d_p*>(pk + PACK128_INV + b * 256 + 2 * t);
  };
  auto pack_deposit = [&](int b, double* img) {
    const int u = b & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < 128 * b) *reinterpret_cast<v2d_p*>(img + (ch >> 3) * 16 + (ch & 7) * 2) = pr[u][q];
    }
    if (t < 128) *reinterpret_cast<v2d_p*>(img + 112 * 16 + 2 * t) = pinv[u];
  };
  // the waves of column half h put their 32 x 64 piece where the slabs' block steps read it
  auto tile_deposit = [&](int h) {
    if (wc == h) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) stage[((2 * wr + ii) * 4 + j) * TL_TS + (g + 4 * q) * 18 + i] = x[ii][j][q];
    }
  };
  pack_request(0);
  pack_request(1);
  tile_deposit(0);
  pack_deposit(0, pkb[0]);
  __syncthreads();
  if (stamps && t == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();
  v4d_t X[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const double* img = pkb[b & 1];
    double* ts = stage + (wave * 4 + (b & 3)) * TL_TS;
    if (b + 2 < 8) pack_request(b + 2);
    v4d_t T;
#pragma unroll
    for (int r = 0; r < 4; ++r) T[r] = ts[i * 18 + g + 4 * r];
#pragma unroll
    for (int a = 0; a < b; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) T = __builtin_amdgcn_mfma_f64_16x16x4f64(-img[(16 * a + g + 4 * r) * 16 + i], X[a][r], T, 0, 0, 0);
    const double* inv = img + 112 * 16;
    X[b] = (v4d_t){0., 0., 0., 0.};
#pragma unroll
    for (int r = 0; r < 4; ++r) X[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(inv[(g + 4 * r) * 16 + i], T[r], X[b], 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) ts[i * 18 + g + 4 * r] = X[b][r];
    __builtin_amdgcn_wave_barrier();
    if (stamps && t == 0) stamps[4 + 2 * b] = __builtin_amdgcn_s_memrealtime();      // X_b of wave 0 is in LDS: its MFMA chain has completed
    {
      const v2d_p o0 = *reinterpret_cast<const v2d_p*>(ts + sr0 * 18 + sp);
      const v2d_p o1 = *reinterpret_cast<const v2d_p*>(ts + (sr0 + 8) * 18 + sp);
      st16<SC1_OUT>(ab, slab + (size_t)sr0 * ld + 16 * b + sp, o0);
      st16<SC1_OUT>(ab, slab + (size_t)(sr0 + 8) * ld + 16 * b + sp, o1);
    }
    if (b == 3) {
      __syncthreads();                // every slab has read block 3 and taken X_3 out of its slot
      if (stamps && t == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
      tile_deposit(1);
    }
    if (b < 7) pack_deposit(b + 1, pkb[(b + 1) & 1]);
    __syncthreads();
    if (stamps && t == 0) stamps[5 + 2 * b] = __builtin_amdgcn_s_memrealtime();      // all four waves are through step b
  }
  if (stamps && t == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace mogp
