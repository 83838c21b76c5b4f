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
//   D(c)      the 128 x 128 diagonal block of block column c (chol128_dev); needs the two diagonal GEMM tasks of column c
//   G(r, c)   r in {2c, 2c+1}: 64 rows of the diagonal block receive the panels 0 .. c-1 (left-looking, long K)
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
#include <cstdlib>
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
constexpr int MC_LDS_HDR = 2;                     // doubles in front of the operand buffers: [task / ok words]

__device__ __forceinline__ unsigned ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stu(unsigned* p, unsigned x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct McCtx {
  unsigned* ctrl;
  int spin_limit;
  int* shi;          // LDS: [0] task number, [1] result of a wait
};

// All 256 threads call.  Lane 0 polls until min(*a, *b, *c) >= want (b, c may equal a); returns that minimum, or -1 after a
// timeout / when another workgroup has aborted.
__device__ __forceinline__ int mc_wait_min3(const McCtx& cx, const unsigned* a, const unsigned* b, const unsigned* c, unsigned want) {
  if (threadIdx.x == 0) {
    int res, spins = 0;
    for (;;) {
      unsigned m = ldu(a);
      const unsigned mb = ldu(b), mc = ldu(c);
      m = m < mb ? m : mb;
      m = m < mc ? m : mc;
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
  }
  __syncthreads();
  const int r = cx.shi[1];
  __syncthreads();
  return r;
}

}  // namespace

// table[p] = (type << 30) | (c << 15) | r;  type 0: D(c), 1: G(r, c), 2: T(r, c)
__global__ __launch_bounds__(256, 2) void mchol_kernel(BatchView v, unsigned* __restrict__ ctrl, const int* __restrict__ table, int ntasks,
                                                       int emu_stride, double* __restrict__ packs, int* __restrict__ info, int nq, int spin_limit) {
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
  for (int qi = 0; qi < nq; ++qi) {
    const int q = (home + qi) & (nq - 1);                 // own queue first, then help the others
    unsigned* head = ctrl + MC_HEADS + q * MC_LINE;
    for (;;) {
      if (t == 0) shi[0] = (int)__hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int tk = shi[0];
      __syncthreads();
      if (tk >= total) break;
      const int p = tk / emus_q, zl = tk - p * emus_q;
      const int z = zl * nq + q;
      const int emu = __builtin_amdgcn_readfirstlane(v.idx ? v.idx[z] : z);
      const int word = table[p];
      const int type = (word >> 30) & 3, c = (word >> 15) & 0x7fff, r = word & 0x7fff;
      double* A = v.A + (size_t)emu * v.MS;
      unsigned* rowdone = ctrl + MC_EMU0 + (size_t)emu * emu_stride;
      unsigned* diagcnt = rowdone + K2;
      unsigned* ddone = diagcnt + K;
      double* pk = packs + ((size_t)emu * K + c) * PACK128_STRIDE;
      const int c0 = 128 * c;
      if (type == 0) {
        // ---- D(c): diagonal block ------------------------------------------------------------------------------------
        if (c > 0 && mc_wait_min3(cx, diagcnt + c, diagcnt + c, diagcnt + c, 2u) < 0) return;
        chol128_dev<true>(A + (size_t)c0 * ld + c0, ld, pk, info + emu, c0, lds);
        __builtin_amdgcn_s_setprio(0);
        drain_stores();
        __syncthreads();
        if (t == 0) stu(ddone + c, 1u);
        continue;
      }
      // ---- G / T: 64 rows x 128 columns receive the panels 0 .. c-1 -----------------------------------------------------
      const int r0 = 64 * r;
      if (c > 0) {
        v4d acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0., 0., 0., 0.};
        int kb = 0;
        while (kb < c) {
          int m = mc_wait_min3(cx, rowdone + r, rowdone + 2 * c, rowdone + 2 * c + 1, (unsigned)(kb + 1));
          if (m < 0) return;
          m = m < c ? m : c;
          mainloop_w<64, 128, 2, 2, false, false>(A + (size_t)r0 * ld + 128 * kb, ld, A + (size_t)c0 * ld + 128 * kb, ld, 8 * (m - kb), acc, lds);
          kb = m;
        }
        if (type == 1) {
          // read by D(c) on another CU: write-through
          for_each_acc_w<2>(acc, [&](int row, int col, double x) {
            double* pc = A + (size_t)(r0 + row) * ld + (c0 + col);
            __hip_atomic_store(pc, *pc - x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          });
        } else {
          for_each_acc_w<2>(acc, [&](int row, int col, double x) {
            double* pc = A + (size_t)(r0 + row) * ld + (c0 + col);
            *pc -= x;
          });
        }
        drain_stores();
        __syncthreads();
      }
      if (type == 1) {
        if (t == 0) __hip_atomic_fetch_add(diagcnt + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      // ---- T: panel solve with the pack of D(c) ---------------------------------------------------------------------------
      if (mc_wait_min3(cx, ddone + c, ddone + c, ddone + c, 1u) < 0) return;
      trsm128_lds_dev<true>(v, c0, r0, pk, emu, 0, lds);
      drain_stores();
      __syncthreads();
      if (t == 0) stu(rowdone + r, (unsigned)(c + 1));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
// Topological task order of ONE emulator.  Column c: D(c); the two row blocks of the NEXT diagonal block first -- T(2c+2, c),
// T(2c+3, c) -- and right behind them the diagonal GEMM tasks of column c+1 that wait for exactly those two, so that the
// chain D(c) -> T -> G -> D(c+1) is taken by workgroups that are already polling when their input arrives; then the rest of
// column c from the top down.
std::vector<int> mchol_task_table(int NP) {
  const int K = NP / 128, K2 = NP / 64;
  std::vector<int> tb;
  auto word = [](int type, int c, int r) { return (type << 30) | (c << 15) | r; };
  for (int c = 0; c < K; ++c) {
    tb.push_back(word(0, c, 0));
    for (int r = 2 * c + 2; r < std::min(2 * c + 4, K2); ++r) tb.push_back(word(2, c, r));
    if (c + 1 < K) {
      tb.push_back(word(1, c + 1, 2 * c + 2));
      tb.push_back(word(1, c + 1, 2 * c + 3));
    }
    for (int r = 2 * c + 4; r < K2; ++r) tb.push_back(word(2, c, r));
  }
  return tb;
}

int mchol_emu_stride(int NP) { return (NP / 64 + 2 * (NP / 128) + MC_LINE - 1) / MC_LINE * MC_LINE; }
size_t mchol_ctrl_ints(int NP, int B) { return MC_EMU0 + (size_t)B * mchol_emu_stride(NP); }
size_t mchol_pack_doubles(int NP, int B) { return (size_t)B * (NP / 128) * PACK128_STRIDE; }

void launch_mchol(const BatchView& v, unsigned* ctrl, size_t ctrl_ints, const int* table, int ntasks, double* packs, int* info, int n_cu,
                  hipStream_t s) {
  // MOGP_MC_SPIN: polls before a wait gives up (default 2^22: seconds); MOGP_MC_WGS: workgroups per CU (default 2)
  static const int spin_limit = [] { const char* e = getenv("MOGP_MC_SPIN"); return e ? atoi(e) : (1 << 22); }();
  static const int per_cu = [] { const char* e = getenv("MOGP_MC_WGS"); return e ? std::max(1, atoi(e)) : 2; }();
  (void)hipMemsetAsync(ctrl, 0, ctrl_ints * sizeof(unsigned), s);
  const int nq = (v.nb % 8 == 0) ? 8 : 1;
  const size_t lds_doubles = MC_LDS_HDR + std::max<size_t>({(size_t)WCfg<64, 128, 2, 2>::SMEM_DOUBLES, (size_t)TRSM128L_LDS, (size_t)C128_LDS_DOUBLES});
  const int total = ntasks * v.nb;
  const int grid = std::min(per_cu * n_cu, total);
  prof_begin("mchol", s);
  hipLaunchKernelGGL(mchol_kernel, dim3(grid), dim3(256), lds_doubles * sizeof(double), s, v, ctrl, table, ntasks, mchol_emu_stride(v.NP), packs, info,
                     nq, spin_limit);
  const double n = v.NP;
  prof_end("mchol", s, (double)v.nb * n * n * n / 3.0, (double)v.nb * 8.0 * n * n);
}

}  // namespace mogp
