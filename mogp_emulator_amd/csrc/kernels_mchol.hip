// The blocked Cholesky of a batch as ONE launch: a dependency-ordered task queue worked off by persistent workgroups.
// Replaces, like the multi-launch schedules of Engine::factorize_blocked, cusolverDnDpotrf (densegp_gpu.hpp:451-474) /
// LAPACK dpotrf (linalg/cholesky.py:225-232).
//
// Why.  With few matrices per GPU (the 8-emulator shard of the 8-GPU run, 2 x n=5000, one n=16000 matrix) the multi-launch
// schedules are bound by the dependent chain of one block column -- short update -> 128 x 128 diagonal block -> panel solve,
// three launches of 25 + 50 + 10 us plus 6-11 us at every kernel boundary and event wait, ~95 us x NP/128 -- not by the
// matrix cores (profiles/r02b_timeline_lookahead_B8.txt).  Here the same three device functions (gemm_dev.h, chol128_dev.h,
// trsm_dev.h) run as TASKS of one kernel, and a task waits for exactly the tiles it needs:
//
//   D(c)      the 128 x 128 diagonal block of block column c (chol128_dev): applies panel c-1 to the register-resident block
//             (chol128_dev<.., PRE>) and factors it; needs T(2c, c-1), T(2c+1, c-1) and the two diagonal GEMM tasks of column c
//   G(s, c)   s = 0, 1, 2: the lower 64 x 64 tile (0,0), (1,0), (1,1) of the diagonal block receives the panels 0 .. c-2 (left-looking,
//             long K; off the chain)
//   T(r, c)   r >= 2c+2: 64 rows x 128 columns below the diagonal block: the same long-K GEMM, then the panel solve with the
//             pack of D(c) (trsm128_lds_dev).  The accumulators stay in registers while the task waits for panel c-1, so the
//             look-ahead of the stream schedules (their U1 / U2 split and its extra read-modify-write pass) is implicit:
//             only the LAST 128 columns of K sit in the dependent chain.
//
// Tasks are numbered per emulator in a TOPOLOGICAL order (every task only depends on tasks with a smaller number) and the
// numbers of all emulators are interleaved into one queue (eight queues -- one per XCD, emulator z in queue z mod 8 -- when
// the batch is a multiple of 8, so that an emulator's tiles stay in one L2).  A workgroup takes the next number with one
// atomic add and runs the task.  Forward progress therefore needs NO assumption about dispatch order or residency: the task
// with the smallest number among the running ones never waits for anything that is not finished or running.  Every wait is
// bounded all the same; a timeout sets the abort word, every workgroup leaves, and the engine factorises the batch again
// with the multi-launch schedule (Engine::factorize_blocked).
//
// Hand-offs (MI355X_MICROARCH.md "inter-workgroup visibility"; cdna_hip_programming.md guideline 16, recipe R1): data that
// another workgroup reads -- solved panel rows, the diagonal-block pack, the diagonal tiles after their GEMM -- is stored
// WRITE-THROUGH (sc1), every storing wave drains its stores (inline-asm s_waitcnt vmcnt(0)), a barrier, then ONE lane
// publishes the counter with a relaxed agent-scope atomic; consumers poll that word relaxed from ONE lane.  No consumer
// ever reads an address before its final value has been published (tiles are write-once per launch from a reader's point
// of view), so no cache can hold a stale copy and no acquire invalidation is needed; tiles and packs are 128-byte aligned,
// so no line is shared between tasks.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
#define MOGP_OPAQUE_TID 1
#include "launch.h"
#include "chol128_dev.h"
#include "trsm_dev.h"
#include "gemm_dev.h"

namespace mogp {

namespace {

constexpr int MC_LINE = 32;                       // ints per 128-byte line
constexpr int MC_ABORT = 0, MC_TIMEOUTS = 1;      // ctrl[0], ctrl[1]
constexpr int MC_HEADS = MC_LINE;                 // queue head q at ctrl[MC_HEADS + q * MC_LINE]
constexpr int MC_EMU0 = MC_LINE * 9;              // per-emulator blocks start here
constexpr int MC_AHEAD_BIT = 1 << 29;              // task word: listed in front of the diagonal block it waits for (mchol_task_table)
constexpr int MC_PD = 4;                          // 16-column pieces a GEMM task consumes per call of its main loop (and the steps its global loads run ahead)
constexpr int MC_LDS_HDR = 4;                     // doubles in front of the operand buffers: [task / ok words]

__device__ __forceinline__ unsigned ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stu(unsigned* p, unsigned x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#ifndef MOGP_MC_SCHED
#define MOGP_MC_SCHED 3
#endif
constexpr int MC_SCHED = MOGP_MC_SCHED;
// the GEMM main loop of the tasks (gemm_dev.h): mainloop_q, two global-load steps ahead of its three LDS stages
// (PIN: the step's barrier stays behind its last MFMAs -- probe 2.13 -> 2.08 us per k-step with two workgroups per CU, 8 x n=2000 0.707 -> 0.693 ms,
// level elsewhere; profiles/r05_loop_pin_ab.txt)
template <int BM, int BN>
__device__ __forceinline__ void MC_GEMM(const double* __restrict__ Ag, int lda, const double* __restrict__ Bg, int ldb, int nk, v4d (&acc)[BM / 32][BN / 32],
                                        double* smem, int kmask) {
  mainloop_q<BM, BN, 2, 2, 2, MC_SCHED>(Ag, lda, Bg, ldb, nk, acc, smem, kmask);
}
constexpr size_t MC_GEMM_LDS = QCfg<64, 128>::SMEM_DOUBLES;

struct McCtx {
  unsigned* ctrl;
  int spin_limit;
  int* shi;          // LDS: [0] task number, [1] result of a wait
};

// optional per-task time stamps (tools/mchol_trace.py; MOGP_MC_TRACE=<file>): MC_TRW 64-bit words per task,
// [0] pulled, [1] last operand wait begins, [2] ends (D: its inputs are complete), [3] after the GEMM main loop, [4] after the
// tile has been written back, [5] published, [6] hardware id, [7] task word | queue position << 32, [8] pack seen (T), [9] sum
// of all waits of the task; 100 MHz clock (s_memrealtime)
constexpr int MC_TRW = 30;      // [10..12] inside the panel solve of a bulk task (trsm128_tile_dev), [13] its stores drained, [14 + 2b] / [15 + 2b] block step b: wave 0's chain done / all waves through
template <bool TRACE>
__device__ __forceinline__ void mc_stamp(unsigned long long* tr, int i) {
  if (TRACE && tr && threadIdx.x == 0) tr[i] = __builtin_amdgcn_s_memrealtime();
}

// All 256 threads call.  Lane 0 polls until min(*a, *b, *c) >= want (b, c may equal a); returns that minimum, or -1 after a
// timeout / when another workgroup has aborted.
__device__ __forceinline__ int mc_wait_min4(const McCtx& cx, const unsigned* a, const unsigned* b, const unsigned* c, const unsigned* d, unsigned want,
                                            unsigned long long* waited = nullptr) {
  if (threadIdx.x == 0) {
    const unsigned long long w0 = waited ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int res, spins = 0;
    for (;;) {
      unsigned m = ldu(a);
      const unsigned mb = ldu(b), mc = ldu(c), md = ldu(d);
      m = m < mb ? m : mb;
      m = m < mc ? m : mc;
      m = m < md ? m : md;
      if (m >= want) {
        res = (int)m;
        break;
      }
      ++spins;
      if (spins > cx.spin_limit) {
        stu(cx.ctrl + MC_ABORT, 1u);
        __hip_atomic_fetch_add(cx.ctrl + MC_TIMEOUTS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        res = -1;
        break;
      }
      if ((spins & 31) == 0 && ldu(cx.ctrl + MC_ABORT) != 0u) {
        res = -1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    cx.shi[1] = res;
    if (waited) waited[9] += __builtin_amdgcn_s_memrealtime() - w0;
  }
  __syncthreads();
  const int r = cx.shi[1];
  __syncthreads();
  return r;
}

__device__ __forceinline__ int mc_wait_min3(const McCtx& cx, const unsigned* a, const unsigned* b, const unsigned* c, unsigned want,
                                            unsigned long long* waited = nullptr) {
  return mc_wait_min4(cx, a, b, c, c, want, waited);
}

// chol128_dev<.., PRE>'s view of the panel to its left: piece j (columns 16j .. 16j+15 of both 64-row tiles) is there once both
// progress words have reached base + j + 1.  One poll returns how far the solves are, so a finished panel costs one wait.
struct PieceWait {
  const McCtx& cx;
  const unsigned *a, *b;
  unsigned base;
  int avail;
  unsigned long long* tr;
  __device__ __forceinline__ bool known(int j) const { return avail > j; }
  __device__ __forceinline__ bool operator()(int j) {
    if (avail > j) return true;
    const int m = mc_wait_min3(cx, a, b, b, base + (unsigned)j + 1u, tr);
    if (m < 0) return false;
    avail = m - (int)base;
    return true;
  }
};

}  // namespace

// table[p] = (type << 30) | (c << 15) | r (built with unsigned arithmetic);  type 0: D(c), 1: G(s, c) with s in the r field, 2: T(r, c)
// SOLO: the launch runs ONE workgroup per CU (chain-bound batches): the kernel may then use the whole register file of a SIMD for its one
// wave -- the diagonal block's spills go to AGPRs instead of scratch memory
// (Round 5: the measurement instantiations of rounds 3 - 4 are gone -- PAIRS, the paired 128 x 128 bulk task behind MOGP_MC_PAIR, 5 % slower
// everywhere; LEGACY, the round-3 forms behind MOGP_MC_TILE / MOGP_MC_SLAB / MOGP_MC_CHAINX / MOGP_MC_PIECES = 0.  Their measurements are in
// HISTORY.md; in this one function every extra path costs registers in ALL paths, which is why they had lived in instantiations of
// their own.)
template <bool TRACE, bool SOLO = false>
__global__ __launch_bounds__(256, SOLO ? 1 : 2) void mchol_kernel(BatchView v, unsigned* __restrict__ ctrl, const int* __restrict__ table, int ntasks,
                                                       int emu_stride, double* __restrict__ packs, int* __restrict__ info, int nq, int spin_limit,
                                                       unsigned long long* __restrict__ trace, int tile_solve) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  int* shi = reinterpret_cast<int*>(smem);
  double* lds = smem + MC_LDS_HDR;
  McCtx cx{ctrl, spin_limit, shi};
  const int t = threadIdx.x;
  const int ld = v.LD;
  const int K = v.NP / 128, K2 = v.NP / 64;
  const int home = (int)blockIdx.x & (nq - 1);          // observed: block b runs on XCD b % 8 (for speed only)
  const int emus_q = v.nb / nq;
  const int total = ntasks * emus_q;
  const int gs = ((tile_solve >> 12) & 0xff) ? ((tile_solve >> 12) & 0xff) : emus_q;
  for (int qi = 0; qi < nq; ++qi) {
    const int q = (home + qi) & (nq - 1);                 // own queue first, then help the others
    unsigned* head = ctrl + MC_HEADS + q * MC_LINE;
    // The number of the NEXT task is drawn while the current one runs (one atomic round trip per task off the path) -- but
    // late in the current task, behind its long GEMM phase: a number drawn at the start sat unserved for the whole task, and
    // when it belonged to the dependent chain a block column waited 12 - 45 us for it (per-task stamps).  Holding one number
    // ahead keeps the progress argument: a held number is larger than the holder's current one, so the smallest current task
    // among all workgroups still depends on finished or running tasks only.
    int next_tk = -1;
    auto draw_next = [&]() {
      if (t == 0 && next_tk < 0) next_tk = (int)__hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    for (;;) {
      draw_next();
      if (t == 0) shi[0] = next_tk;
      __syncthreads();
      const int tk = shi[0];
      __syncthreads();
      if (tk >= total) break;
      next_tk = -1;
      // tickets: groups of gs emulators of this queue, one group's tasks (interleaved over its emulators) before the next group's
      const int per = ntasks * gs, grp = tk / per, rem = tk - grp * per;
      const int p = rem / gs, zl = grp * gs + (rem - p * gs);
      const int z = zl * nq + q;
      const int emu = __builtin_amdgcn_readfirstlane(v.idx ? v.idx[z] : z);
      const int word = table[p];
      const int type = (word >> 30) & 3, c = (word >> 15) & 0x3fff, r = word & 0x7fff;
      const bool listed_ahead = (word & MC_AHEAD_BIT) != 0;     // (stands in front of the D it waits for: draws no ticket while it waits)
      double* A = v.A + (size_t)emu * v.MS;
      unsigned* rowdone = ctrl + MC_EMU0 + (size_t)z * emu_stride;      // control rows and packs belong to the batch SLOT z of this launch
      unsigned* diagcnt = rowdone + K2;
      unsigned* ddone = diagcnt + K;
      unsigned* rowprog = ddone + K;          // row tile r: 8 c + b = the first b 16-column pieces of its block column c are visible
      double* pk = packs + ((size_t)z * K + c) * PACK128_STRIDE;
      const int c0 = 128 * c;
      unsigned long long* tr = TRACE ? trace + ((size_t)z * ntasks + p) * MC_TRW : nullptr;
      if (TRACE && t == 0) {
        tr[6] = __builtin_amdgcn_s_getreg(6164) | ((unsigned long long)__builtin_amdgcn_s_getreg(((8 - 1) << 11) | (8 << 6) | 4) << 8);   // XCC_ID, HW_ID
        tr[7] = (unsigned)word | ((unsigned long long)tk << 32);
      }
      mc_stamp<TRACE>(tr, 0);
      if (type == 0) {
        // ---- D(c): diagonal block ------------------------------------------------------------------------------------
        mc_stamp<TRACE>(tr, 1);
        // the panels 0 .. c-2 arrive through the two G tasks (c >= 2), panel c-1 is applied here, straight from the two
        // panel-solve tasks that produced it
        if (c > 1 && mc_wait_min3(cx, diagcnt + c, diagcnt + c, diagcnt + c, 3u, tr) < 0) return;
        auto inputs_seen = [&]() { mc_stamp<TRACE>(tr, 2); };
        // (ddone[c] counts the block steps whose pack entries are visible: 8 = the whole pack)
        if (c > 0) {
          // panel c-1 is consumed in 16-column pieces while the two panel-solve tasks still produce it (rowprog)
          PieceWait pw{cx, rowprog + 2 * c, rowprog + 2 * c + 1, 8u * (unsigned)(c - 1), 0, tr};
          if (!chol128_dev<true, true, true>(A + (size_t)c0 * ld + c0, ld, pk, info + emu, c0, lds, ddone + c, pw, inputs_seen)) return;
        } else {
          inputs_seen();
          chol128_dev<true, false, true>(A + (size_t)c0 * ld + c0, ld, pk, info + emu, c0, lds, ddone + c);
        }
        __builtin_amdgcn_s_setprio(0);
        drain_stores();
        __syncthreads();
        if (t == 0) stu(ddone + c, 8u);
        mc_stamp<TRACE>(tr, 5);
        continue;
      }
      if (type == 1) {
        // ---- G(s, c): lower 64 x 64 tile s = (0,0), (1,0), (1,1) of the diagonal block receives the panels 0 .. c-2 -----------
        // (three 64 x 64 tasks rather than two 64 x 128 ones: no work on the upper-right quarter, and half the length per task -- a
        // 64 x 128 task of a late block column needed ~7 c us of a whole CU, more than one block-column period, and the diagonal
        // block waited for it: 65 - 98 us periods in the per-task stamps)
        const int ti = r > 0 ? 1 : 0, tj = r > 1 ? 1 : 0;
        const int gi0 = c0 + 64 * ti, gj0 = c0 + 64 * tj;
        __builtin_amdgcn_s_setprio(2);
        v4d acc[2][2];              // (SOLO: starts as -C, see the T tasks)
        double* pcG = A + (size_t)(gi0 + (t >> 7) * 32 + ((t & 63) >> 4)) * ld + (gj0 + ((t >> 6) & 1) * 32 + (t & 15));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = SOLO ? -pcG[(size_t)(i * 16 + 4 * q) * ld + j * 16] : 0.0;
        const int kend = c - 1;
        int kb = 0;
        while (kb < kend) {
          mc_stamp<TRACE>(tr, 1);
          int m = mc_wait_min3(cx, rowdone + 2 * c + ti, rowdone + 2 * c + tj, rowdone + 2 * c + tj, (unsigned)(kb + 1), tr);
          if (m < 0) return;
          mc_stamp<TRACE>(tr, 2);
          m = m < kend ? m : kend;
          MC_GEMM<64, 64>(A + (size_t)gi0 * ld + 128 * kb, ld, A + (size_t)gj0 * ld + 128 * kb, ld, 8 * (m - kb), acc, lds, -1);
          kb = m;
        }
        mc_stamp<TRACE>(tr, 3);
        draw_next();
        {
          double cv[2][2][4];
          if (!SOLO) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) cv[i][j][q] = pcG[(size_t)(i * 16 + 4 * q) * ld + j * 16];
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q)      // read by D(c) on another CU: write-through
                __hip_atomic_store(pcG + (size_t)(i * 16 + 4 * q) * ld + j * 16, SOLO ? -acc[i][j][q] : cv[i][j][q] - acc[i][j][q], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        drain_stores();
        mc_stamp<TRACE>(tr, 4);
        __syncthreads();
        if (t == 0) __hip_atomic_fetch_add(diagcnt + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mc_stamp<TRACE>(tr, 5);
        __builtin_amdgcn_s_setprio(0);
        continue;
      }
      // ---- T: 64 rows x 128 columns receive the panels 0 .. c-1 -----------------------------------------------------
      const int r0 = 64 * r;
      // tasks of the dependent chain (diagonal tiles, the two row blocks of the next diagonal block) issue ahead of the
      // workgroup they share the CU with
      // (tile_solve bits 8, 9: 4 + 2 x rows below the diagonal block are chain tasks -- chain-bound launches take three pairs)
      const bool urgent = r < 2 * c + 4 + 2 * ((tile_solve >> 8) & 3);
      if (urgent) __builtin_amdgcn_s_setprio(2);
      const int kend = c;
      if (kend > 0) {
        // SOLO (chain-bound launches, one workgroup per CU): acc starts as -C -- the tile's covariance entries are requested with the first
        // operands instead of after the GEMM, where their latency sits on the dependent chain (one n = 2000 matrix: mchol 0.471 -> 0.454 ms);
        // the solve then takes x = -acc.  Throughput-bound launches read C after the GEMM as before: there the early read costs 1 % (its
        // latency is covered by the CU partner anyway, and the first MFMAs wait for it).  Round 5, profiles/r05_negc_ab.txt.
        v4d acc[2][4];
        const double* pcT = A + (size_t)(r0 + (t >> 7) * 32 + ((t & 63) >> 4)) * ld + (c0 + ((t >> 6) & 1) * 64 + (t & 15));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = SOLO ? -pcT[(size_t)(i * 16 + 4 * q) * ld + j * 16] : 0.0;
        // x = C - (sum of products), whichever way acc started
        auto finish_x = [&]() {
          if (SOLO) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i][j] = -acc[i][j];
          } else {
            double cv[2][4][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) cv[i][j][q] = pcT[(size_t)(i * 16 + 4 * q) * ld + j * 16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][j][q] = cv[i][j][q] - acc[i][j][q];
          }
        };
        {
          // The k range is consumed in 16-column PIECES as they become visible (rowprog counts them: whole block columns of finished
          // tiles plus what a chain task has published of the tile it is solving), MC_PD pieces at a time.  Why: every tile of block
          // column c needs the rows of the diagonal block -- the chain tasks of column c-1 -- for its last 128 columns; waiting for those
          // tiles to be complete put the whole 8-step GEMM (13 us on one CU) behind them, in the dependent chain of the next column
          // (per-task stamps of one n = 2000 matrix: operands 1.2 us after the chain task, GEMM 13.4, solve 16.4 = the 34 us period).
          int ks = 0;
          const int kse = 8 * kend;
          while (ks < kse) {
            mc_stamp<TRACE>(tr, 1);
            int have = mc_wait_min3(cx, rowprog + r, rowprog + 2 * c, rowprog + 2 * c + 1, (unsigned)(ks + MC_PD), tr);
            if (have < 0) return;
            mc_stamp<TRACE>(tr, 2);
            have = (have < kse ? have : kse) & ~(MC_PD - 1);
            MC_GEMM<64, 128>(A + (size_t)r0 * ld + 16 * ks, ld, A + (size_t)c0 * ld + 16 * ks, ld, have - ks, acc, lds,
                             (tile_solve & 2) ? 3 : -1);
            ks = have;
          }
        }
        mc_stamp<TRACE>(tr, 3);
        if (!listed_ahead) draw_next();
        if (!urgent) {
          // bulk task, round 4: the tile is re-dealt to the solving waves through LDS BEFORE the solve (trsm128_tile2_dev); the first
          // pack images are requested before the C tile is read
          if (mc_wait_min3(cx, ddone + c, ddone + c, ddone + c, 8u, tr) < 0) return;       // (a formality for a bulk task)
          TrsmSlabPre SP;
          trsm128_slab_request(pk, SP);
          finish_x();
          mc_stamp<TRACE>(tr, 4);
          mc_stamp<TRACE>(tr, 8);
          __builtin_amdgcn_s_setprio(1);
          trsm128_tile2_dev<true>(v, c0, r0, pk, emu, lds, acc, SP, TRACE ? tr + 10 : nullptr);
          drain_stores();
          mc_stamp<TRACE>(tr, 13);
          __syncthreads();
          if (t == 0) {
            stu(rowprog + r, 8u * (unsigned)(c + 1));
            stu(rowdone + r, (unsigned)(c + 1));
          }
          mc_stamp<TRACE>(tr, 5);
          __builtin_amdgcn_s_setprio(0);
          continue;
        }
        {
          // chain task, round 4: x = C - acc stays in registers and is re-dealt to the solving waves through LDS (trsm128_tile2_chain_dev)
          // instead of being written back and re-read by the pipelined solve
          finish_x();
          mc_stamp<TRACE>(tr, 4);
          bool first = true;
          auto wait = [&](int b) {
            const bool ok = mc_wait_min3(cx, ddone + c, ddone + c, ddone + c, (unsigned)(b + 1), tr) >= 0;
            if (first) mc_stamp<TRACE>(tr, 8);
            first = false;
            return ok;
          };
          auto pub = [&](int b) {
            if (t == 0) stu(rowprog + r, 8u * (unsigned)c + (unsigned)b + 1u);
          };
          const int prog = mc_wait_min3(cx, ddone + c, ddone + c, ddone + c, 0u);
          if (prog < 0) return;
          if (prog >= 8) {
            mc_stamp<TRACE>(tr, 8);
            trsm128_tile2_chain_dev<true, true>(v, c0, r0, pk, emu, lds, acc, TrsmNoWait(), pub);
          } else if (!trsm128_tile2_chain_dev<true, false>(v, c0, r0, pk, emu, lds, acc, wait, pub)) return;
          drain_stores();
          __syncthreads();
          if (t == 0) {
            stu(rowprog + r, 8u * (unsigned)(c + 1));
            stu(rowdone + r, (unsigned)(c + 1));
          }
          mc_stamp<TRACE>(tr, 5);
          __builtin_amdgcn_s_setprio(0);
          continue;
        }
      }
      // (only block column 0 gets here: its tiles have no GEMM and are solved straight from the covariance entries in A)
      // ---- T: panel solve with the pack of D(c) ---------------------------------------------------------------------------
      if (urgent) {
        // chain task: one block step behind the diagonal block that is still being factored (chol128_dev<.., PROG>)
        bool first = true;
        auto wait = [&](int b) {
          const bool ok = mc_wait_min3(cx, ddone + c, ddone + c, ddone + c, (unsigned)(b + 1), tr) >= 0;
          if (first) mc_stamp<TRACE>(tr, 8);
          first = false;
          return ok;
        };
        // (its solved rows are published piece by piece: the next diagonal block applies them as they come)
        auto pub = [&](int b) {
          if (t == 0) stu(rowprog + r, 8u * (unsigned)c + (unsigned)b + 1u);
        };
        // Late block columns of chain-bound launches: the task's own GEMM ends AFTER the diagonal block has finished (per-task stamps: pack
        // seen 10 - 24 us after D).  The pipelined form then still fetches every row-block image behind its (satisfied) wait, one exposed
        // memory round trip per block step: 15 us per solve against 8 - 9 with the images requested two steps ahead.
        const int prog = mc_wait_min3(cx, ddone + c, ddone + c, ddone + c, 0u);
        if (prog < 0) return;
        if (prog >= 8) {
          mc_stamp<TRACE>(tr, 8);
          trsm128_lds_dev<true, true, false>(v, c0, r0, pk, emu, 0, lds, TrsmNoWait(), pub);
        } else if (!trsm128_lds_dev<true, false, true>(v, c0, r0, pk, emu, 0, lds, wait, pub)) return;
      } else {
        if (mc_wait_min3(cx, ddone + c, ddone + c, ddone + c, 8u, tr) < 0) return;
        mc_stamp<TRACE>(tr, 8);
        // the substitution is a chain of dependent MFMAs: let it issue ahead of the co-resident workgroup's dense MFMA stream
        __builtin_amdgcn_s_setprio(1);
        trsm128_lds_dev<true, true>(v, c0, r0, pk, emu, 0, lds);
      }
      drain_stores();
      __syncthreads();
      if (t == 0) {
        stu(rowprog + r, 8u * (unsigned)(c + 1));
        stu(rowdone + r, (unsigned)(c + 1));
      }
      mc_stamp<TRACE>(tr, 5);
      __builtin_amdgcn_s_setprio(0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
// Topological task order of ONE emulator.  The dependent chain D(c) -> T(2c+2, c), T(2c+3, c) -> D(c+1) is listed ONE BLOCK COLUMN
// AHEAD of the bulk of its column: the queue hands out numbers only as fast as workgroups come free (with 8 x n=2000 on 256
// workgroups it runs barely ahead of the chain), and a chain task drawn 35 us before its diagonal block still had 50 - 70 us of
// long-K GEMM of its own to do (per-task stamps: the block column then took 65 - 100 us instead of 48).  The pair below the
// chain rows, T(2c+4, c), T(2c+5, c), is part of the chain too -- the chain tasks of column c+1 cannot apply panel c to their
// own rows before it is solved -- and sits in front of the bulk of column c-1 (listed behind D(c+1) its 8c k-steps of GEMM
// started when the chain already needed it: in the late block columns of n = 2000 the panel solves saw their pack 15 - 20 us
// after the diagonal block had finished).  Order:
//   D(0); T(2, 0), T(3, 0); T(4, 0), T(5, 0);
//   for c = 0, 1, ...:  D(c+1);  T(2c+4, c+1), T(2c+5, c+1);  G(0..2, c+2);  T(2c+6, c), T(2c+7, c);  T(2c+6, c+1), T(2c+7, c+1);
//                       T(r, c) for r >= 2c+8
// (a workgroup that draws a chain task early does its GEMM and then waits: at most a handful of waiting workgroups per emulator).
// Round 5 measured other orders for the LAST block columns, where a launch has fewer tasks than workgroups and ends on the row band 2c+6, 2c+7
// (per-task stamps: the last tickets of 64 x n=2000 are drawn at 3.5 ms of 3.7): that band at the front of its iteration, or directly behind
// the chain pair: level; the G tasks in front of D(c+1): 2 % slower (profiles/r05_task_order_ab.txt).
// ahead (round 5): the BAND tasks of an iteration -- the chain pair T(2c+4, c+1), T(2c+5, c+1), the pair below it T(2c+6, c+1), T(2c+7, c+1) and
// that pair's tasks of the column before, T(2c+6, c), T(2c+7, c) -- are listed at the END of iteration c-1, directly IN FRONT of the diagonal
// block D(c+1) four of them wait for, and carry MC_AHEAD_BIT.  Drawn a few tickets earlier, they have more of their long-K GEMM behind them
// when that block finishes (8 x n=2000 0.689 -> 0.670 ms, 16 x 1.126 -> 1.078, one matrix 0.451 -> 0.445).  This is a BOUNDED exception to
// "every task only depends on smaller numbers": a flagged task may wait for the D listed at most 6 places behind it, with only flagged tasks
// in between.  Forward progress then needs (a) that a flagged task holds no drawn-ahead ticket while it waits (the kernel draws its next
// number only when it is done: that ticket could be the very D it waits for), and (b) more workgroups per queue than the 7 tickets per
// emulator that can be waiting in front of one D: launch_mchol uses the in-order table otherwise.
std::vector<int> mchol_task_table(int NP, bool ahead) {
  const int K = NP / 128, K2 = NP / 64;
  auto word = [](int type, int c, int r) { return (int)(((unsigned)type << 30) | ((unsigned)c << 15) | (unsigned)r); };
  std::vector<std::pair<long, int>> keyed;
  const long M = 4L * K2 + 64;                       // keys per iteration
  long slot = 0;
  int iter = -1;
  auto put = [&](int w, bool band) {
    const bool moved = band && ahead && iter >= 1;
    keyed.push_back({(long)((moved ? iter - 1 : iter) + 1) * M + (moved ? M / 2 : 0) + slot, moved ? (w | MC_AHEAD_BIT) : w});
    ++slot;
  };
  auto T = [&](int r, int c, bool band) {
    if (c < K && r < K2 && r >= 2 * c + 2) put(word(2, c, r), band);
  };
  put(word(0, 0, 0), false);
  T(2, 0, false);
  T(3, 0, false);
  T(4, 0, false);
  T(5, 0, false);
  for (int c = 0; c < K; ++c) {
    iter = c;
    slot = 0;
    if (c + 1 < K) put(word(0, c + 1, 0), false);
    T(2 * c + 4, c + 1, true);
    T(2 * c + 5, c + 1, true);
    // (three 64 x 64 G tasks per diagonal block; round 5 measured the tiles (1,0), (1,1) as ONE 64 x 128 task for throughput-bound launches:
    // level and bit-identical, profiles/r05_wide_g_ab.txt -- not kept)
    if (c + 2 < K)
      for (int sub = 0; sub < 3; ++sub) put(word(1, c + 2, sub), false);
    T(2 * c + 6, c, true);
    T(2 * c + 7, c, true);
    T(2 * c + 6, c + 1, true);
    T(2 * c + 7, c + 1, true);
    for (int r = 2 * c + 8; r < K2; ++r) T(r, c, false);
  }
  std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<long, int>& a, const std::pair<long, int>& b) { return a.first < b.first; });
  std::vector<int> tb;
  for (auto& kw : keyed) tb.push_back(kw.second);
  return tb;
}

int mchol_emu_stride(int NP) { return (2 * (NP / 64) + 2 * (NP / 128) + MC_LINE - 1) / MC_LINE * MC_LINE; }   // rowdone | diagcnt | ddone | rowprog
size_t mchol_ctrl_ints(int NP, int B) { return MC_EMU0 + (size_t)B * mchol_emu_stride(NP); }
size_t mchol_pack_doubles(int NP, int B) { return (size_t)B * (NP / 128) * PACK128_STRIDE; }

// rho = (time the matrix cores need at ~45 TFLOP/s) / (length of the dependent chain, ~55 us per block column)
static double mchol_rho(int nb, int NP) {
  const double npd = NP;
  return ((double)nb * npd * npd * npd / 3.0 / 45e12) / ((npd / 128.0) * 55e-6);
}

void launch_mchol(const BatchView& v, unsigned* ctrl, size_t ctrl_ints, const int* tables, int ntasks, double* packs, int* info, int n_cu,
                  hipStream_t s, bool ctrl_zeroed) {
  // MOGP_MC_SPIN: polls before a wait gives up (default 2^22: seconds)
  static const int spin_limit = [] { const char* e = getenv("MOGP_MC_SPIN"); return e ? atoi(e) : (1 << 22); }();
  // Workgroups per CU by regime.  rho = (time the matrix cores need at ~45 TFLOP/s) / (length of the dependent chain, ~55 us per block
  // column).  Chain-bound batches run ONE workgroup per CU, so that a diagonal-block task never shares its CU's matrix pipes; beyond that two
  // per CU.  Before the interleaved k-step the crossover was rho = 1 (mchol ms one / two per CU, profiles/r05_regime_sweep.txt: 4 x n=2000
  // 0.48 / 0.58, 8 x 0.71 / 0.73, 12 x 0.92 / 0.91 - 0.93, 16 x 1.17 / 1.15, 24 x 1.68 / 1.56, 32 x 2.13 / 1.97, 2 x n=5000 2.10 / 2.14, 4 x n=5000
  // 3.73 / 3.61); now see below (profiles/r05_regime_sweep2.txt).  MOGP_MC_WGS = 1 / 2 forces either.  (Rounds 2 - 4: for 1 <= rho < 2 the workgroup sharing a CU with a
  // diagonal-block task PARKED, MOGP_MC_PARK; on the round-5 kernels parking is level to 2 % slower in every regime and is gone.)
  static const int force_wgs = [] { const char* e = getenv("MOGP_MC_WGS"); return e ? std::max(1, atoi(e)) : 0; }();
  const double rho = mchol_rho(v.nb, v.NP);
  // (with the interleaved k-step a lone workgroup's GEMM runs at 0.974 us per step, a pair at 1.873 for two: one per CU gives up 4 % of GEMM
  // throughput and keeps the chain free -- it now wins up to 14 x n=2000 (equal at 16; 20 x 1.34 / 1.29, 32 x 2.04 / 1.94, 64 x 3.96 / 3.66),
  // up to 4 x n=5000 (3 x 2.65 / 2.86, 4 x 3.43 / 3.58; equal at 6 - 8; 16 x 12.95 / 12.76) and for one n=16000 matrix (24.39 / 24.80): the
  // threshold grows with the depth of the matrix, whose share of GEMM work it follows)
  const double K16 = std::max(1.0, (v.NP / 128) / 16.0);
  const int per_cu = force_wgs ? force_wgs : (rho < 1.2 * std::pow(K16, 0.7) ? 1 : 2);
  // bit 1: MOGP_MC_NOTRAFFIC=1 (measurement only, garbage results): the bulk GEMM tasks re-read their first 64 operand columns -- the traffic A/B
  static const int tile_solve = [] {
    const char* f = getenv("MOGP_MC_NOTRAFFIC");
    return (f && atoi(f)) ? 2 : 0;
  }();
  // MOGP_MC_URG = 0 / 1 / 2: two, four or six row tiles below the diagonal block are chain tasks (pipelined solve, pieces published)
  static const int force_urg = [] { const char* e = getenv("MOGP_MC_URG"); return e ? atoi(e) & 3 : -1; }();

  if (!ctrl_zeroed) (void)hipMemsetAsync(ctrl, 0, ctrl_ints * sizeof(unsigned), s);
  const int nq = (v.nb % 8 == 0) ? 8 : 1;
  // one workgroup per CU is enforced through the LDS request: more than half of the 160 KB
  const size_t lds_need = MC_LDS_HDR + std::max<size_t>({(size_t)MC_GEMM_LDS, (size_t)TRSM128L_LDS, (size_t)C128_LDS_PRE_DOUBLES, (size_t)TRSM128T_LDS});
  const size_t lds_doubles = per_cu == 1 ? std::max<size_t>(lds_need, 10 * 1024 + 64) : lds_need;
  const int total = ntasks * v.nb;
  const int grid = std::min(per_cu * n_cu, total);
  // the band-ahead order (mchol_task_table) needs more workgroups per queue than tickets that can wait in front of one diagonal block;
  // MOGP_MC_AHEAD=0: always the in-order table
  static const bool ahead_on = [] { const char* e = getenv("MOGP_MC_AHEAD"); return !e || atoi(e) != 0; }();
  // (measured, mchol ms in-order / band-ahead: 4 x n=2000 0.485 / 0.477, 8 x 0.700 / 0.676, 12 x 0.923 / 0.905, 16 x 1.122 / 1.106, 24 x 1.556 / 1.534,
  // 32 x level, 64 x 3.69 / 3.77, one matrix 0.460 / 0.463, n=5000 and n=16000 level: profiles/r05_band_ahead_ab.txt -- so: 0.2 <= rho < 2)
  const bool ahead = ahead_on && rho >= 0.2 && rho < 2.0 && 8 * (v.nb / nq) <= grid / nq;
  const int* table = tables + (ahead ? ntasks : 0);
  // MOGP_MC_TRACE=<file>: per-task time stamps of EVERY launch are appended to the file (analysis only: synchronises)
  static const char* trace_file = getenv("MOGP_MC_TRACE");
  const size_t words = (size_t)total * MC_TRW;
  unsigned long long* dtr = nullptr;
  // Emulators per ticket GROUP inside a queue (round 5): a queue hands out the tasks of gs of its emulators, interleaved, before the next gs -- the
  // next group starts in the tail of the one before, and fewer matrices are in flight per XCD.  The smallest divisor of the queue's emulators
  // that still offers 1.25 x as many row tiles as the queue has workgroups.  mchol ms, all / groups: 64 x n=2000 (8 per queue) 3.54 / 3.47 in
  // fours (twos 3.61, ones 3.98), 16 x n=5000 (2 per queue) 12.48 / 12.32 in ones; bit-identical.  MOGP_MC_EGRP: 0 = no groups, n = groups of n.
  static const int egrp = [] { const char* e = getenv("MOGP_MC_EGRP"); return e ? atoi(e) : -1; }();
  const int emus_q = v.nb / nq, wg_q = std::max(1, grid / nq);
  int gsz = 0;
  if (egrp > 0) gsz = (emus_q % egrp == 0) ? egrp : 0;
  else if (egrp < 0 && v.NP >= 2048)           // (128 x n=1000, sixteen per queue: groups of eight 1.222 against 1.198 ms undivided -- short matrices stay undivided;
                                               //  96 x n=2000 5.43 -> 5.20, 120 x 6.77 -> 6.40, 32 x n=5000 24.9 -> 24.5)
    for (int g = 1; g < emus_q; ++g)
      if (emus_q % g == 0 && 4 * g * (v.NP / 64) >= 5 * wg_q) {
        gsz = g;
        break;
      }
  const int ts = tile_solve | ((force_urg >= 0 ? force_urg : (rho < 1.0 ? 2 : 0)) << 8) | (gsz << 12);
  if (trace_file) {
    if (hipMalloc(reinterpret_cast<void**>(&dtr), words * 8) != hipSuccess) {
      dtr = nullptr;                                        // no room for the stamps: factorise untraced, and say so
      fprintf(stderr, "libmogp_hip: MOGP_MC_TRACE: no device memory for %zu stamp words, this factorisation is not traced\n", words);
    }
  }
  if (trace_file && dtr) {
    (void)hipMemsetAsync(dtr, 0, words * 8, s);
    hipLaunchKernelGGL(mchol_kernel<true>, dim3(grid), dim3(256), lds_doubles * sizeof(double), s, v, ctrl, table, ntasks, mchol_emu_stride(v.NP), packs,
                       info, nq, spin_limit, dtr, ts);
    std::vector<unsigned long long> h(words);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), dtr, words * 8, hipMemcpyDeviceToHost);
    (void)hipFree(dtr);
    if (FILE* f = fopen(trace_file, "ab")) {
      const long long hdr[4] = {v.nb, ntasks, v.NP, grid + 1000000LL * MC_TRW};
      fwrite(hdr, sizeof(hdr), 1, f);
      fwrite(h.data(), 8, words, f);
      fclose(f);
    }
    return;
  }
  prof_begin("mchol", s);
  static const int solo_ok = [] { const char* e = getenv("MOGP_MC_SOLO"); return e ? atoi(e) : 1; }();
  if (per_cu == 1 && solo_ok)
    hipLaunchKernelGGL((mchol_kernel<false, true>), dim3(grid), dim3(256), lds_doubles * sizeof(double), s, v, ctrl, table, ntasks, mchol_emu_stride(v.NP), packs,
                     info, nq, spin_limit, (unsigned long long*)nullptr, ts);
  else
    hipLaunchKernelGGL(mchol_kernel<false>, dim3(grid), dim3(256), lds_doubles * sizeof(double), s, v, ctrl, table, ntasks, mchol_emu_stride(v.NP), packs,
                     info, nq, spin_limit, (unsigned long long*)nullptr, ts);
  const double n = v.n;                 // ALGORITHMIC work (n^3 / 3 per emulator), not the padded NP the tiles cover
  prof_end("mchol", s, (double)v.nb * n * n * n / 3.0, (double)v.nb * 8.0 * n * n);
}

}  // namespace mogp
