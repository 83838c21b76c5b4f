// Log-determinant, triangular solves, the triangular-inverse leaf and small consumers of the factor.  These replace
// cusolverDnDpotrs as used by the reference (densegp_gpu.hpp:576-591) and the cub log-diagonal reduction
// (util.cu:38-49).  The factorisation kernels themselves live in kernels_gemm.hip / chol128_dev.h / trsm_dev.h.
#include <algorithm>
#include <cstdlib>
#include "launch.h"
#include "chol128_dev.h"

namespace mogp {

typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int slot_emu(const int* idx, int z) { return idx ? idx[z] : z; }

// ---------------------------------------------------------------------------------------------
// logdet = 2 sum_{i<n} log L_ii ;  yty = sum_{c<n} L[n,c]^2   (row n of the factor holds y^T)
// ---------------------------------------------------------------------------------------------
// res[emu * RES_STRIDE + ..]: [0] log-determinant, [1] factorisation status word (as a double), [2 + r * RMAX + c] Gram matrix --
// everything the host needs after a factorisation in ONE device-to-host copy (three separate copies cost ~20 us each).
// A back substitution that gave up waiting (backsolve_chain_kernel) is reported as status BACKSOLVE_TIMEOUT -- a code of its
// own, never confused with a failed factorisation (> 0): the engine repeats the solve of that emulator with the multi-launch path.
// The sums of one emulator by one 256-thread workgroup: res[0] and the Gram matrix (everything but the status word).  Also called by the
// leftmost chunk of the one-launch back substitution (round 5), which has nothing to do until the chain reaches it: same code, same bits.
// (RM: right-hand-side rows handled -- RMAX in logdet_kernel, 1 in the back substitution, whose launches have one; an entry's sum does not depend on RM)
template <int RM>
__device__ __forceinline__ void logdet_gram_dev(const BatchView& v, int emu, double* __restrict__ res, double* red) {
  const int ld = v.LD, R = v.R;
  const double* A = v.A + (size_t)emu * v.MS;
  double s = 0.;
  double acc[RM * (RM + 1) / 2];
#pragma unroll
  for (int e = 0; e < RM * (RM + 1) / 2; ++e) acc[e] = 0.;
  for (int i = threadIdx.x; i < v.n; i += 256) {
    s += log(A[(size_t)i * ld + i]);
    double y[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) y[r] = (r < R) ? A[(size_t)(v.n + r) * ld + i] : 0.0;
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) acc[r * (r + 1) / 2 + c] = __builtin_fma(y[r], y[c], acc[r * (r + 1) / 2 + c]);
  }
  auto block_sum = [&](double x) {
    red[threadIdx.x] = x;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    const double out = red[0];
    __syncthreads();
    return out;
  };
  const double ls = block_sum(s);
  double* out = res + (size_t)emu * RES_STRIDE;
  if (threadIdx.x == 0) out[0] = 2.0 * ls;
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      if (r < R) {                    // uniform
        const double g = block_sum(acc[r * (r + 1) / 2 + c]);
        if (threadIdx.x == 0) {
          out[2 + r * RMAX + c] = g;
          out[2 + c * RMAX + r] = g;
        }
      }
    }
}

__global__ __launch_bounds__(256) void logdet_kernel(BatchView v, const int* __restrict__ info, double* __restrict__ res,
                                                     const int* __restrict__ bs_status, int bs_epoch, const unsigned* __restrict__ mc_abort) {
  __shared__ double red[256];
  const int emu = slot_emu(v.idx, blockIdx.x);
  logdet_gram_dev<RMAX>(v, emu, res, red);
  if (threadIdx.x == 0) {
    const int st = info[emu];
    int rep = (st == 0 && bs_status && bs_status[emu] == bs_epoch) ? BACKSOLVE_TIMEOUT : st;
    if (mc_abort && *mc_abort != 0u) rep = MCHOL_ABORTED;       // the one-launch Cholesky gave up: nothing in A is usable
    res[(size_t)emu * RES_STRIDE + 1] = (double)rep;
  }
}

// Multi-launch back substitution alpha = L^-T y (several right-hand sides, and the fall-back of the one-launch chain below):
// per 64-row block one tiny diagonal-solve launch (one wave per emulator) and one gemv launch spread over as many
// workgroups as there are column chunks, so that all CUs stream L (one workgroup per emulator is bound by ONE CU's
// streaming rate, ~25 GB/s: 0.7 ms at n=2000, 40 ms at n=16000; that variant was removed in round 3).
__global__ __launch_bounds__(256) void backsolve_init_kernel(BatchView v) {
  const int emu = slot_emu(v.idx, blockIdx.y);
  const int ld = v.LD, n = v.n, R = v.R;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ld) return;
  for (int r = 0; r < R; ++r)
    v.Z[((size_t)emu * R + r) * ld + i] = (i < n) ? v.A[(size_t)emu * v.MS + (size_t)(n + r) * ld + i] : 0.0;
}

// one wave: x = L_kk^-T w[k0 .. k0+64) for a single right-hand side (lane t holds column t of L_kk)
__device__ __forceinline__ void backsolve_diag1_wave(const BatchView& v, int emu, int k0) {
  const int ld = v.LD, n = v.n;
  const double* A = v.A + (size_t)emu * v.MS;
  const int t = threadIdx.x & 63;
  double u[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) u[j] = A[(size_t)(k0 + j) * ld + k0 + t];
  const double rdg = 1.0 / A[(size_t)(k0 + t) * ld + k0 + t];
  double* w = v.Z + (size_t)emu * ld;
  double b = w[k0 + t];
  double xout = 0.0;
#pragma unroll
  for (int j = 63; j >= 0; --j) {
    double xj = readlane_f64(b * rdg, j);
    if (k0 + j >= n) xj = 0.0;
    if (t == j) xout = xj;
    b = __builtin_fma(-u[j], xj, b);
  }
  w[k0 + t] = xout;
}

__global__ __launch_bounds__(64) void backsolve_diag_kernel(BatchView v, int k0) {
  const int emu = slot_emu(v.idx, blockIdx.x);
  const int ld = v.LD, n = v.n, R = v.R;
  const double* A = v.A + (size_t)emu * v.MS;
  const int t = threadIdx.x;
  double u[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) u[j] = A[(size_t)(k0 + j) * ld + k0 + t];
  const double rdg = 1.0 / A[(size_t)(k0 + t) * ld + k0 + t];
  for (int r = 0; r < R; ++r) {
    double* w = v.Z + ((size_t)emu * R + r) * ld;
    double b = w[k0 + t];
    double xout = 0.0;
#pragma unroll
    for (int j = 63; j >= 0; --j) {
      double xj = readlane_f64(b * rdg, j);
      if (k0 + j >= n) xj = 0.0;
      if (t == j) xout = xj;
      b = __builtin_fma(-u[j], xj, b);
    }
    w[k0 + t] = xout;
  }
}

// w_r[c] -= sum_i L[k0+i][c] alpha_r[k0+i] for c < k0 and every right-hand side r; one column pair per thread,
// L is read once for all right-hand sides
constexpr int BSG_THREADS = 64;     // small workgroups: at n=16000, B=1 a 256-thread version has only 32 workgroups in flight
template <int RT>     // RT: compile-time bound on the number of right-hand sides (1 = plain fit path)
__global__ __launch_bounds__(BSG_THREADS) void backsolve_gemv_kernel(BatchView v, int k0) {
  __shared__ double ab[RT][64];
  const int emu = slot_emu(v.idx, blockIdx.y);
  const int ld = v.LD, R = v.R;
  const double* A = v.A + (size_t)emu * v.MS;
  double* w = v.Z + (size_t)emu * R * ld;
  for (int e = threadIdx.x; e < R * 64; e += BSG_THREADS) ab[e >> 6][e & 63] = w[(size_t)(e >> 6) * ld + k0 + (e & 63)];
  __syncthreads();
  const int c = 2 * (blockIdx.x * BSG_THREADS + threadIdx.x);
  if (c >= k0) return;
  v2d s[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) s[r] = (v2d){0., 0.};
  const double* p = A + (size_t)k0 * ld + c;
#pragma clang loop unroll_count(RT == 1 ? 32 : 8)
  for (int i = 0; i < 64; ++i) {
    const v2d x = *reinterpret_cast<const v2d*>(p + (size_t)i * ld);
#pragma unroll
    for (int r = 0; r < RT; ++r)
      if (RT == 1 || r < R) {
        const double ar = ab[r][i];
        s[r][0] = __builtin_fma(x[0], ar, s[r][0]);
        s[r][1] = __builtin_fma(x[1], ar, s[r][1]);
      }
  }
#pragma unroll
  for (int r = 0; r < RT; ++r)
    if (RT == 1 || r < R) {
      v2d cur = *reinterpret_cast<v2d*>(w + (size_t)r * ld + c);
      cur[0] -= s[r][0];
      cur[1] -= s[r][1];
      *reinterpret_cast<v2d*>(w + (size_t)r * ld + c) = cur;
    }
}

// single right-hand side, 256 threads: the 64 rows of the block are split over the four waves (16 loads in flight per
// thread instead of two batches of 32), partial sums are combined in a fixed order through LDS
// FUSE_DIAG: the workgroup that owns the columns of the NEXT diagonal block [k0-64, k0) finishes them here and solves that
// block right away (one wave), so the separate diagonal-solve launch of the next step disappears
template <bool FUSE_DIAG>
__global__ __launch_bounds__(256) void backsolve_gemv4_kernel(BatchView v, int k0) {
  __shared__ double ab[64];
  __shared__ v2d part[3][64];
  const int emu = slot_emu(v.idx, blockIdx.y);
  const int ld = v.LD;
  const double* A = v.A + (size_t)emu * v.MS;
  double* w = v.Z + (size_t)emu * ld;
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  if (threadIdx.x < 64) ab[threadIdx.x] = w[k0 + threadIdx.x];
  __syncthreads();
  const int c = 2 * (blockIdx.x * 64 + lane);
  v2d s = {0., 0.};
  if (c < k0) {
    const double* p = A + (size_t)(k0 + 16 * rg) * ld + c;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const v2d x = *reinterpret_cast<const v2d*>(p + (size_t)i * ld);
      const double ar = ab[16 * rg + i];
      s[0] = __builtin_fma(x[0], ar, s[0]);
      s[1] = __builtin_fma(x[1], ar, s[1]);
    }
  }
  if (rg > 0) part[rg - 1][lane] = s;
  __syncthreads();
  if (rg == 0 && c < k0) {
    s += part[0][lane];
    s += part[1][lane];
    s += part[2][lane];
    v2d cur = *reinterpret_cast<v2d*>(w + c);
    cur -= s;
    *reinterpret_cast<v2d*>(w + c) = cur;
  }
  if (FUSE_DIAG && (int)blockIdx.x == (k0 - 64) / 128) {      // workgroup-uniform
    __syncthreads();
    if (threadIdx.x < 64) backsolve_diag1_wave(v, emu, k0 - 64);
  }
}

// ---------------------------------------------------------------------------------------------
// alpha = L^-T y in ONE launch (single right-hand side): a dependency-ordered chain of workgroups.  Workgroup (emulator,
// chunk c) owns the 128 entries [128c, 128c+128) of the solution.  It folds in every 64-row block of L below its own
// rows as soon as the owner of those rows has published its part of alpha -- the tile of L is already in registers by
// then, loads do not depend on alpha -- then solves its own two 64 x 64 diagonal blocks (one wave, columns of the block
// held in LDS) and publishes.  Chunks are dispatched right to left (the rightmost chunk has no dependencies), so a
// workgroup only ever waits for workgroups with a smaller block index: no deadlock however many are resident.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, valid form "8-byte agent atomics on both sides";
// cdna_hip_programming.md guideline 16, recipe R1): the payload is written with agent-scope atomic stores (sc1: through to
// L2), EVERY storing wave drains them (inline-asm s_waitcnt: the compiler may drop a builtin wait in front of a flag store),
// a barrier, then one lane stores the flag; the consumer polls the flag relaxed and reads the payload with agent-scope
// atomic loads (sc1: past its own L1), so no cached copy of another workgroup's alpha is ever read.
// Forward progress is NOT assumed: in-order dispatch makes a wait short, but HIP does not promise it, so every wait is
// bounded and a workgroup that gives up records status[emu] = epoch (a word of its own -- not the factorisation's info,
// which would start the jitter ladder) and still publishes, so nobody behind it hangs; the engine then repeats the solve
// of that emulator with the multi-launch path (Engine::eval, after_factor).
// Replaces 32 launches of ~6 us each at n = 2000 (0.2 ms of a 1.7 ms fit for 8 emulators, 0.43 of 5.3 ms for 64).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// SENT (round 4, the default; MOGP_BS_SENTINEL=0 keeps the flag form): no flag and no second round trip -- the solution vector was
// preset to all-ones bit patterns by the K build, a chunk's entries are published by their own agent-scope stores, and a consumer's
// lanes poll the 128 VALUES they need until they are no longer that pattern.  Per chain step this removes the producer's drain +
// barrier + flag store and the consumer's payload load behind its flag poll (~2.5 of 7 us), and the lower half of a chunk (solved
// first) is folded while its producer still solves the upper half.  A lane that gives up stores the emulator's status word, takes 0.0 (never the
// pattern, which an fma would propagate into its own results and make every chunk behind it time out too) and the emulator is re-solved by the engine.
// HOIST (launches with at most one workgroup per CU -- the chain-bound ones; needs SENT): see preload_diag below; the one-per-CU build has the
// registers for it (with two per CU the 128 extra live registers spilled into the solve they were meant to shorten).
template <bool SENT, bool HOIST = false>
__global__ __launch_bounds__(256, HOIST ? 1 : 2) void backsolve_chain_kernel(BatchView v, int* __restrict__ flags, int epoch, int nch, int* __restrict__ status, int spin_limit,
                                                                 const int* __restrict__ info, double* __restrict__ res, const unsigned* __restrict__ mc_abort) {
  __shared__ double Ld[2][64 * 65];        // the two diagonal blocks of this chunk: [row][column], row stride 65
  __shared__ double w[128], xs[128];
  __shared__ v2d part[3][64];
  __shared__ int timed_out;
  __shared__ double red[256];
  // (dispatch order: every emulator's rightmost chunk first.  Round 5 measured groups of 32 / 16 / 8 / 4 / 1 emulators, each with all its chunks
  // before the next group's, for launches with more workgroups than the GPU holds -- 64 x 16 on 512 slots, where the leftmost chunks,
  // which stream the most rows of L, become resident last: level, profiles/r05_backsolve_order_ab.txt)
  const int cp = blockIdx.x / v.nb, z = blockIdx.x % v.nb;
  const int c = nch - 1 - cp;              // rightmost chunk first in dispatch order
  const int emu = slot_emu(v.idx, z);
  const int ld = v.LD, n = v.n;
  const double* A = v.A + (size_t)emu * v.MS;
  double* alpha = v.Z + (size_t)emu * ld;
  int* fl = flags + (size_t)emu * nch;
  const int t = threadIdx.x, lane = t & 63, rg = __builtin_amdgcn_readfirstlane(t >> 6);      // (wave-uniform: row addresses are scalar)
  const int j0 = 128 * c;
  if (t == 0) timed_out = 0;
  if (t < 128) w[t] = (j0 + t < n) ? A[(size_t)n * ld + j0 + t] : 0.0;
  // the chunk's diagonal blocks, full 512-byte rows
  for (int e = t; e < 2 * 64 * 32; e += 256) {
    const int blk = e >> 11, r = (e >> 5) & 63, cc = (e & 31) * 2;
    const int row = j0 + 64 * blk + r;
    v2d x = {0., 0.};
    if (row < n) x = *reinterpret_cast<const v2d*>(A + (size_t)row * ld + j0 + 64 * blk + cc);
    Ld[blk][r * 65 + cc] = x[0];
    Ld[blk][r * 65 + cc + 1] = x[1];
  }
  // tile of 64 rows [k0, k0+64) x columns [j0, j0+128): lane -> two columns, wave -> 16 of the rows.  TWO register sets: both tiles of
  // the chunk that is folded next (and, last, the chunk's own off-diagonal tile) are requested a whole step before they are used --
  // the loads do not depend on alpha, so their latency sits under the wait for the flag instead of on the chain (with one set the
  // second tile of every chunk and the own tile were requested when they were needed: ~1.5 us each per chain step)
  v2d tA[16], tB[16];
  auto load_tile = [&](v2d (&tv)[16], int k0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = k0 + 16 * rg + i;
      // (SENT: unpredicated -- rows n .. NP-1 exist, hold finite values (forward-solved targets, identity padding) and meet x = 0; with the
      // predicate every row became a basic block of its own in that form and the register allocator spilled 500 bytes per lane)
      if (SENT) tv[i] = *reinterpret_cast<const v2d*>(A + (size_t)row * ld + j0 + 2 * lane);
      else tv[i] = (row < n) ? *reinterpret_cast<const v2d*>(A + (size_t)row * ld + j0 + 2 * lane) : (v2d){0., 0.};
    }
  };
  // w[cols] -= tile^T x, x = xs[xoff .. xoff+64); ncols = 128 (a block below the chunk) or 64 (inside the chunk)
  auto apply_tile = [&](const v2d (&tv)[16], int xoff, int ncols) {
    v2d s = {0., 0.};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double ar = xs[xoff + 16 * rg + i];
      s[0] = __builtin_fma(tv[i][0], ar, s[0]);
      s[1] = __builtin_fma(tv[i][1], ar, s[1]);
    }
    if (rg > 0) part[rg - 1][lane] = s;
    __syncthreads();
    if (rg == 0 && 2 * lane < ncols) {
      s += part[0][lane];
      s += part[1][lane];
      s += part[2][lane];
      w[2 * lane] -= s[0];
      w[2 * lane + 1] -= s[1];
    }
    __syncthreads();
  };
  // SENT: the lane's 64 scaled column entries of a diagonal block, -L[j][lane] / L_ll, in REGISTERS, held by the wave that solves the block: wave 0
  // the lower block (rows j0+64.., solved first), wave 1 the upper one.  HOIST (round 5): fetched at the START of the kernel, before the chunk's
  // dependencies arrive (they depend on L only); otherwise inside each solve -- 64 LDS reads + 64 multiplies in front of the 64-step substitution,
  // ~1 of the 2.3 us of a block, twice per step of the launch's dependent chain.
  double Lc[64], rdgw = 0.0;
  const int myblk = rg == 0 ? 1 : 0;           // (waves 2, 3: unused)
  auto preload_diag = [&]() {
    if (SENT && rg < 2) {
      const double* Lb = Ld[myblk];
      const double dg = Lb[lane * 65 + lane];
      rdgw = (j0 + 64 * myblk + lane < n) ? 1.0 / dg : 0.0;
#pragma unroll
      for (int j = 0; j < 64; ++j) Lc[j] = Lb[j * 65 + lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 64; ++j) Lc[j] *= -rdgw;
    }
  };
  // x = L_kk^-T w[woff .. woff+64) for diagonal block blk (one wave; lane t holds column t of the block), into xs and alpha
  auto solve_diag = [&](int blk) {
    const int k0 = j0 + 64 * blk;
    if (SENT ? rg == (blk ? 0 : 1) : rg == 0) {
      const double* Lb = Ld[blk];
      double xout = 0.0;
      if (SENT) {
        if (!HOIST) preload_diag();
        // the lane carries b / L_ll instead of b: per step readlane -> fma (the multiply by 1 / L_ll left the chain; the scaled column
        // entries L[j][lane] / L_ll do not depend on the right-hand side)
        double bs = w[64 * blk + lane] * rdgw;
        __builtin_amdgcn_sched_barrier(0);
        int xlo = 0, xhi = 0;
#pragma unroll
        for (int j = 63; j >= 0; --j) {
          // x_j: lane j's value, to every lane through two scalar registers -- and back into lane j of the result with v_writelane
          // (a compare + selects per step were five of the eight instructions of a step)
          const int lo = __builtin_amdgcn_readlane(__double2loint(bs), j), hi = __builtin_amdgcn_readlane(__double2hiint(bs), j);
          asm("v_writelane_b32 %0, %1, %2" : "+v"(xlo) : "s"(lo), "n"(j));
          asm("v_writelane_b32 %0, %1, %2" : "+v"(xhi) : "s"(hi), "n"(j));
          bs = __builtin_fma(Lc[j], __hiloint2double(hi, lo), bs);      // rows >= n: rdg = 0 -> x_j = 0 (identity padding, right-hand-side rows)
        }
        xout = __hiloint2double(xhi, xlo);
      } else {
      const double dg = Lb[lane * 65 + lane];
      const double rdg = (k0 + lane < n) ? 1.0 / dg : 0.0;
      double b = w[64 * blk + lane];
#pragma unroll
      for (int j = 63; j >= 0; --j) {
        const double xj = readlane_f64(b * rdg, j);       // rows >= n: rdg = 0 -> xj = 0 (identity padding, right-hand-side rows)
        if (lane == j) xout = xj;
        b = __builtin_fma(-Lb[j * 65 + lane], xj, b);      // L[k0+j][k0+lane]; only lanes < j use it afterwards
      }
      }
      xs[64 * blk + lane] = xout;
      if (SENT && __double_as_longlong(xout) == -1ll) xout = __builtin_nan("");      // (only garbage can be the "not there yet" pattern)
      st_agent(alpha + k0 + lane, xout);
    }
    __syncthreads();
  };
  // SENT: lanes [lo, lo + 64) of the workgroup fetch entries lo .. lo+63 of chunk cc into xs, each polling its own value
  auto poll_values = [&](int cc, int lo) {
    if (t >= lo && t < lo + 64) {
      const double* p = alpha + 128 * cc + t;
      double x = ld_agent(p);
      int spins = 0;
      while (__double_as_longlong(x) == -1ll && spins++ < spin_limit) {
        __builtin_amdgcn_s_sleep(1);
        x = ld_agent(p);
      }
      if (__double_as_longlong(x) == -1ll) {
        // Round 6: the time-out has a channel of its own -- the emulator's status word, stored and RELEASED here, before this
        // workgroup publishes anything (its solves sit behind the barrier below), so that the leftmost chunk, which has polled every
        // chunk's values when it reports, is certain to see it.  The lane takes 0.0 (never the pattern).  Round 5 poisoned the value with
        // a NaN instead and let the leftmost chunk infer "timed out" from a NaN in its own result: a legitimately non-finite alpha
        // (inf / NaN targets, overflowed hyper-parameters) was then reported as a time-out, counted and re-solved.
        timed_out = 1;
        __hip_atomic_store(status + emu, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        x = 0.0;
      }
      xs[t] = x;
    }
    __syncthreads();
  };
  __syncthreads();
  if (HOIST) preload_diag();
  // res != nullptr (round 5): the leftmost chunk -- last in the chain, idle until it is reached -- also forms the emulator's log-determinant and
  // Gram entry (the work of logdet_kernel: one launch and its 6 us gap less per evaluation) and, at its end, the status word
  if (SENT && res && c == 0) logdet_gram_dev<1>(v, emu, res, red);
  // blocks below the chunk, from the bottom up: chunk cc' = nch-1 .. c+1, each with two 64-row blocks
  if (nch - 1 > c) {
    load_tile(tA, 128 * (nch - 1) + 64);
    load_tile(tB, 128 * (nch - 1));
  } else {
    load_tile(tA, j0 + 64);                    // rightmost chunk: only its own off-diagonal tile
  }
  for (int cc = nch - 1; cc > c; --cc) {
    if (SENT) {
      poll_values(cc, 64);                     // the lower half of chunk cc is solved (and stored) first
      apply_tile(tA, 64, 128);                 // rows 128cc+64 ..
      if (cc - 1 > c) load_tile(tA, 128 * (cc - 1) + 64);
      else load_tile(tA, j0 + 64);
      poll_values(cc, 0);
      apply_tile(tB, 0, 128);                  // rows 128cc ..
      if (cc - 1 > c) load_tile(tB, 128 * (cc - 1));
      continue;
    }
    if (t == 0) {
      int spins = 0;
      bool seen = false;
      while (!(seen = __hip_atomic_load(fl + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) && spins++ < spin_limit) __builtin_amdgcn_s_sleep(2);
      if (!seen) timed_out = 1;
    }
    __syncthreads();
    if (t < 128) xs[t] = ld_agent(alpha + 128 * cc + t);
    __syncthreads();
    apply_tile(tA, 64, 128);                   // rows 128cc+64 ..
    if (cc - 1 > c) load_tile(tA, 128 * (cc - 1) + 64);
    else load_tile(tA, j0 + 64);               // own rows j0+64 .. x columns j0 .. (only the first 64 columns are used)
    apply_tile(tB, 0, 128);                    // rows 128cc ..
    if (cc - 1 > c) load_tile(tB, 128 * (cc - 1));
  }
  // own rows: upper diagonal block, the 64 x 64 block between the two, lower diagonal block
  solve_diag(1);
  apply_tile(tA, 64, 64);
  solve_diag(0);
  if (SENT) {                                  // (the stores of solve_diag are the publication)
    if (res && c == 0 && t == 0) {
      // every chunk's values have been polled by this one: a wait that gave up anywhere in the chain stored the emulator's status word
      // before that chunk published (poll_values), so it is visible here.  A NaN in the result is NOT a time-out: it surfaces with the
      // factorisation's own status (non-finite log-posterior -> ok = 0) and is not re-solved.
      const int st = info[emu];
      const bool gave_up = timed_out || __hip_atomic_load(status + emu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
      int rep = (st == 0 && gave_up) ? BACKSOLVE_TIMEOUT : st;
      if (mc_abort && *mc_abort != 0u) rep = MCHOL_ABORTED;
      res[(size_t)emu * RES_STRIDE + 1] = (double)rep;
    }
    return;
  }
  // publish: payload drained, then the flag
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    __hip_atomic_store(fl + c, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (timed_out) __hip_atomic_store(status + emu, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// alpha[c] = sum_r M[emu][c][r] Z[r]  (R > 1: Kinv_t_mean and the rank-correction rows from the raw solves)
__global__ __launch_bounds__(256) void combine_rows_kernel(BatchView v, const double* __restrict__ M) {
  const int emu = slot_emu(v.idx, blockIdx.y);
  const int ld = v.LD, R = v.R;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ld) return;
  const double* Zb = v.Z + (size_t)emu * R * ld;
  const double* Mb = M + (size_t)emu * (RMAX + 1) * RMAX;
  double z[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) z[r] = (r < R) ? Zb[(size_t)r * ld + i] : 0.0;
  for (int c = 0; c < v.RA; ++c) {
    double s = 0.;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) s = __builtin_fma(Mb[c * RMAX + r], z[r], s);
    v.alpha[((size_t)emu * v.RA + c) * ld + i] = s;
  }
}

// alpha = Linv^T y (fit+gradient path, Linv already available): alpha_i = sum_{i<=k<n} Linv[k][i] y_k.
// One workgroup per 64-column strip; the 4 waves split the k range, rows are 512-byte coalesced reads.
__global__ __launch_bounds__(256) void alpha_linv_kernel(BatchView v) {
  // a workgroup takes 128 columns of L^-1, a lane two of them (16-byte loads: 8-byte accesses run at 0.54 - 0.70 of that rate on this
  // part), a wave every fourth row; sixteen loads in flight per lane
  __shared__ double red[4][128];
  typedef double a2 __attribute__((ext_vector_type(2)));
  const int emu = slot_emu(v.idx, blockIdx.y);
  const int ld = v.LD, n = v.n, R = v.R;
  const double* Li = v.Linv + (size_t)emu * v.MS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.x * 128, i = i0 + 2 * lane;
  for (int r = 0; r < R; ++r) {
    const double* y = v.A + (size_t)emu * v.MS + (size_t)(n + r) * ld;
    a2 s = {0., 0.};
    if (i0 < n) {
#pragma unroll 16
      for (int k = i0 + wave; k < n; k += 4) {                 // Linv[k][i] = 0 for k < i
        const a2 l = *reinterpret_cast<const a2*>(Li + (size_t)k * ld + i);
        const double yk = y[k];
        s[0] = __builtin_fma(l[0], yk, s[0]);
        s[1] = __builtin_fma(l[1], yk, s[1]);
      }
    }
    red[wave][2 * lane] = s[0];
    red[wave][2 * lane + 1] = s[1];
    __syncthreads();
    if (threadIdx.x < 128) {
      const int c = i0 + (int)threadIdx.x, t = threadIdx.x;
      v.Z[((size_t)emu * R + r) * ld + c] = (c < n) ? red[0][t] + red[1][t] + red[2][t] + red[3][t] : 0.0;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Leave-one-out predictive variance at every training input: 1 / [K^-1]_ii with
// [K^-1]_ii = sum_{k >= i} Linv[k][i]^2.  This is what MICEFastGP.fast_predict obtains per point from
// a Woodbury downdate of the full inverse (SequentialDesign.py:705-747); here all n values come from
// one pass over L^-1.  Same wave split as alpha_linv_kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loo_variance_kernel(BatchView v, double* __restrict__ out, int out_ld) {
  __shared__ double red[4][64];
  const int emu = slot_emu(v.idx, blockIdx.y);
  const int ld = v.LD, n = v.n;
  const double* Li = v.Linv + (size_t)emu * v.MS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.x * 64, i = i0 + lane;
  double s = 0.;
#pragma unroll 8
  for (int k = i0 + wave; k < n; k += 4) {
    const double x = Li[(size_t)k * ld + i];
    s = __builtin_fma(x, x, s);
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && i < n) out[(size_t)blockIdx.y * out_ld + i] = fmax(1.0 / (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]), 0.0);
}

// ---------------------------------------------------------------------------------------------
// History-matching implausibility fused behind the batched prediction (HistoryMatching.py:262-276):
//   I_k(x) = |z_k - mu_k(x)| / sqrt(var_k(x) + nugget_k + discrepancy_k + obsvar_k),
// score(x) = the (rank+1)-th largest I_k(x) over the outputs.  One thread per query point keeps the
// rank+1 largest values in registers; means / variances never leave HBM.
//   prm: per slot [z, obsvar + discrepancy (+ nugget), mean offset]
// ---------------------------------------------------------------------------------------------
constexpr int IMPL_MAXRANK = 16;
__global__ __launch_bounds__(256) void implausibility_kernel(int nb, const double* __restrict__ mean, const double* __restrict__ var, int ld,
                                                             int m, const double* __restrict__ prm, int rank, double* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  double top[IMPL_MAXRANK];
#pragma unroll
  for (int r = 0; r < IMPL_MAXRANK; ++r) top[r] = -1.0;
  for (int k = 0; k < nb; ++k) {
    const double mu = mean[(size_t)k * ld + j] + prm[3 * k + 2];
    const double vv = fmax(var[(size_t)k * ld + j], 0.0) + prm[3 * k + 1];
    double I = fabs(prm[3 * k] - mu) / sqrt(vv);
    // insert into the descending list of the rank+1 largest
#pragma unroll
    for (int r = 0; r < IMPL_MAXRANK; ++r)
      if (r <= rank && I > top[r]) {
        const double t = top[r];
        top[r] = I;
        I = t;
      }
  }
  double res = top[0];
#pragma unroll
  for (int r = 1; r < IMPL_MAXRANK; ++r)
    if (r == rank) res = top[r];
  out[j] = res;
}

// ---------------------------------------------------------------------------------------------
// Mean-function terms of a batched prediction, on the device (densegp_gpu.hpp:334-337, 402-405, 443-447; analytic mean:
// GaussianProcess.py:906-935), so that predictions of models with a mean function can stay in HBM (the sharded predict gathers
// them there) and the host path is the same code followed by one copy.
//   basis (nbasis, m): row 0 = 1, row t = x[dims[t-1]]^powers[t-1] (std::pow on the host: the library's one definition of the
//   basis);  coef (nb, nbasis): the emulator's mean parameters (fixed mean: the value), or beta with an analytic mean.
//   R == 1:  mean[k][j] += coef_0 + sum_t coef_t basis_t[j]
//   R  > 1:  dots (nb, R, m) = [k*^T K^-1 (t - H beta); k*^T K^-1 h_c]:  mean = dots_0 + sum_c beta_c basis_c,
//            var += || LA^-1 (basis[:, j] - dots_{1..q}[:, j]) ||^2   (LA (nb, q, q) lower, row-major)
//   dbasis (nterm, m) = powers[t] x^(powers[t]-1), ddims[t] = its input dimension: deriv[k][j][ddims[t]] += coef_{t+1} dbasis_t[j]
// Products and sums are NOT contracted, in the order of the host loops this replaces (hostmath.h MeanFunc).
// ---------------------------------------------------------------------------------------------
constexpr int MEAN_MAXQ = RMAX;
__device__ __forceinline__ double add_nc(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ double sub_nc(double a, double b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ double mul_nc(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__global__ __launch_bounds__(256) void predict_mean_finish_kernel(int nb, int m, int D, int R, int nbasis, const double* __restrict__ basis,
                                                                  const double* __restrict__ coef, const double* __restrict__ dots,
                                                                  const double* __restrict__ LA, double* __restrict__ mean,
                                                                  double* __restrict__ var, long ld, int nterm,
                                                                  const double* __restrict__ dbasis, const int* __restrict__ ddims,
                                                                  const int* __restrict__ dpowers, double* __restrict__ deriv) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  if (j >= m) return;
  const double* c = coef + (size_t)k * nbasis;
  if (mean) {
    if (R == 1) {
      double v = c[0];
      for (int t = 1; t < nbasis; ++t) v = add_nc(v, mul_nc(c[t], basis[(size_t)t * m + j]));
      mean[(size_t)k * ld + j] = add_nc(mean[(size_t)k * ld + j], v);
    } else {
      const int q = R - 1;
      const double* dk = dots + (size_t)k * R * m;
      double mu = dk[j];
      for (int t = 0; t < q; ++t) mu = add_nc(mu, mul_nc(c[t], basis[(size_t)t * m + j]));
      mean[(size_t)k * ld + j] = mu;
      if (var) {
        const double* La = LA + (size_t)k * q * q;
        double rm[MEAN_MAXQ], add = 0.;
#pragma unroll
        for (int t = 0; t < MEAN_MAXQ; ++t) {
          if (t < q) {
            double sres = sub_nc(basis[(size_t)t * m + j], dk[(size_t)(1 + t) * m + j]);
#pragma unroll
            for (int p = 0; p < MEAN_MAXQ; ++p)
              if (p < t) sres = sub_nc(sres, mul_nc(La[t * q + p], rm[p]));
            rm[t] = sres / La[t * q + t];
            add = add_nc(add, mul_nc(rm[t], rm[t]));
          }
        }
        var[(size_t)k * ld + j] = add_nc(var[(size_t)k * ld + j], add);
      }
    }
  }
  if (deriv && nterm > 0) {
    // terms that share an input dimension are summed first (in term order), then added -- the host's mean_inputderiv
    for (int t = 0; t < nterm; ++t) {
      const int d = ddims[t];
      bool first = true;
      for (int u = 0; u < t; ++u) first = first && (ddims[u] != d);
      if (!first) continue;
      double mid = 0.;
      for (int u = t; u < nterm; ++u)
        if (ddims[u] == d) mid = add_nc(mid, mul_nc(mul_nc(c[u + 1], (double)dpowers[u]), dbasis[(size_t)u * m + j]));
      double* o = deriv + ((size_t)k * m + j) * D + d;
      *o = add_nc(*o, mid);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// trtri leaf: invert every 64x64 diagonal block of L; lane j produces column j of the inverse.
// Also zeroes the block to the right inside the same 128-tile so that 128-granular consumers can
// treat diagonal tiles of Linv as dense.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void trtri_leaf_kernel(BatchView v, const double* __restrict__ Lmat) {
  // the 64 x 64 block goes through LDS: coalesced 512-byte rows in, and every L[c][p] of the substitution is then ONE
  // broadcast LDS read for the whole wave (as scalar loads -- 2016 s_load per wave, returned out of order -- the kernel
  // took 290 us for 64 emulators x n = 2000)
  __shared__ double Ls[64 * 65];
  const int emu = __builtin_amdgcn_readfirstlane(slot_emu(v.idx, blockIdx.y));
  const int ld = v.LD;
  const int d0 = blockIdx.x * 64;
  const double* __restrict__ L = Lmat + (size_t)emu * v.MS + (size_t)d0 * ld + d0;
  double* Li = v.Linv + (size_t)emu * v.MS;
  const int t = threadIdx.x;
#pragma unroll 8
  for (int r = 0; r < 64; ++r) Ls[r * 65 + t] = L[(size_t)r * ld + t];
  __builtin_amdgcn_wave_barrier();
  // lane t computes column t of the inverse: L z = e_t, row-oriented with 4 partial sums
  double x[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    double s0 = (c == t) ? 1.0 : 0.0, s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
    for (int p = 0; p < c; ++p) {
      const double lv = Ls[c * 65 + p];
      if ((p & 3) == 0) s0 = __builtin_fma(-x[p], lv, s0);
      else if ((p & 3) == 1) s1 = __builtin_fma(-x[p], lv, s1);
      else if ((p & 3) == 2) s2 = __builtin_fma(-x[p], lv, s2);
      else s3 = __builtin_fma(-x[p], lv, s3);
    }
    x[c] = ((s0 + s1) + (s2 + s3)) / Ls[c * 65 + c];
  }
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
void launch_logdet(const BatchView& v, const int* info, double* res, hipStream_t s, const int* bs_status, int bs_epoch, const unsigned* mc_abort) {
  hipLaunchKernelGGL(logdet_kernel, dim3(v.nb), dim3(256), 0, s, v, info, res, bs_status, bs_epoch, mc_abort);
}

void launch_combine_rows(const BatchView& v, const double* M, hipStream_t s) {
  hipLaunchKernelGGL(combine_rows_kernel, dim3((v.LD + 255) / 256, v.nb), dim3(256), 0, s, v, M);
}

bool launch_backsolve_chain(const BatchView& v, int* flags, int epoch, int* status, int n_cu, hipStream_t s, const int* info, double* res, const unsigned* mc_abort) {
  const int nch = (v.n + 127) / 128;
  // MOGP_BS_SPIN: polls before a wait gives up (default 2^20, about a second); 0 makes every unsatisfied wait a timeout,
  // which is how the GPU suite exercises the fallback
  static const int spin_limit = [] { const char* e = getenv("MOGP_BS_SPIN"); return e ? atoi(e) : (1 << 20); }();
  static const int sent = [] { const char* e = getenv("MOGP_BS_SENTINEL"); return e ? atoi(e) : 1; }();
  prof_begin("backsolve", s);
  // MOGP_BS_HOIST=0: the chain-bound launches (at most one workgroup per CU) also run the two-per-CU build
  static const int hoist = [] { const char* e = getenv("MOGP_BS_HOIST"); return e ? atoi(e) : 1; }();
  // MOGP_BS_LOGDET=0: log-determinant and status by logdet_kernel behind the chain, as before round 5
  static const int fuse = [] { const char* e = getenv("MOGP_BS_LOGDET"); return e ? atoi(e) : 1; }();
  // (chain-bound launches only: with more workgroups than CUs the leftmost chunks, which stream the most rows of L, are the launch's stragglers)
  double* r = (sent && fuse && v.R == 1 && v.nb * nch <= n_cu) ? res : nullptr;
  if (sent && hoist && v.nb * nch <= n_cu)
    hipLaunchKernelGGL((backsolve_chain_kernel<true, true>), dim3(v.nb * nch), dim3(256), 0, s, v, flags, epoch, nch, status, spin_limit, info, r, mc_abort);
  else if (sent) hipLaunchKernelGGL(backsolve_chain_kernel<true>, dim3(v.nb * nch), dim3(256), 0, s, v, flags, epoch, nch, status, spin_limit, info, r, mc_abort);
  else hipLaunchKernelGGL(backsolve_chain_kernel<false>, dim3(v.nb * nch), dim3(256), 0, s, v, flags, epoch, nch, status, spin_limit, info, r, mc_abort);
  prof_end("backsolve", s, 0., (double)v.nb * 4.0 * (double)v.n * (double)v.n);      // algorithmic: the lower triangle of L read once
  return r != nullptr;
}

void launch_backsolve(const BatchView& v, hipStream_t s) {
  hipLaunchKernelGGL(backsolve_init_kernel, dim3((v.NP + 255) / 256, v.nb), dim3(256), 0, s, v);
  const int nblk = (v.n + 63) / 64;
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int k0 = kb * 64;
    // one right-hand side: the four-wave gemv also solves the NEXT diagonal block (one launch per block instead of two)
    const bool fused = v.R == 1;
    if (!fused || kb == nblk - 1) hipLaunchKernelGGL(backsolve_diag_kernel, dim3(v.nb), dim3(64), 0, s, v, k0);
    if (k0 > 0) {
      const dim3 grid((k0 / 2 + BSG_THREADS - 1) / BSG_THREADS, v.nb);
      if (fused) hipLaunchKernelGGL(backsolve_gemv4_kernel<true>, grid, dim3(256), 0, s, v, k0);
      else hipLaunchKernelGGL(backsolve_gemv_kernel<RMAX>, grid, dim3(BSG_THREADS), 0, s, v, k0);
    }
  }
}

void launch_alpha_from_linv(const BatchView& v, hipStream_t s) {
  hipLaunchKernelGGL(alpha_linv_kernel, dim3(v.NP / 128, v.nb), dim3(256), 0, s, v);
}

void launch_loo_variance(const BatchView& v, double* out, int out_ld, hipStream_t s) {
  hipLaunchKernelGGL(loo_variance_kernel, dim3((v.n + 63) / 64, v.nb), dim3(256), 0, s, v, out, out_ld);
}

void launch_implausibility(int nb, const double* mean, const double* var, int ld, int m, const double* prm, int rank, double* out,
                           hipStream_t s) {
  hipLaunchKernelGGL(implausibility_kernel, dim3((m + 255) / 256), dim3(256), 0, s, nb, mean, var, ld, m, prm, rank, out);
}

// basis (1 + nterm, m) and dbasis (nterm, m) of a polynomial mean function at device-resident test points (round 6, ADVICE r5: the host used to
// evaluate them -- a read-back of the test points, std::pow over m x nterm and a staged upload on the device-resident predict path):
// basis[0] = 1, basis[t + 1][j] = x_j[dims[t]] ^ powers[t], dbasis[t][j] = x_j[dims[t]] ^ (powers[t] - 1)  (hostmath.h MeanFunc).  Powers 0, 1, 2
// are formed exactly (what a correctly rounded pow returns); others through pow.
__device__ __forceinline__ double int_pow(double x, int p) {
  if (p == 0) return 1.0;
  if (p == 1) return x;
  if (p == 2) return x * x;
  return pow(x, (double)p);
}
__global__ __launch_bounds__(256) void mean_basis_kernel(const double* __restrict__ Xs, int m, int D, int nterm, const int* __restrict__ dims,
                                                         const int* __restrict__ powers, double* __restrict__ basis, double* __restrict__ dbasis) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  basis[j] = 1.0;
  for (int t = 0; t < nterm; ++t) {
    const double x = Xs[(size_t)j * D + dims[t]];
    basis[(size_t)(t + 1) * m + j] = int_pow(x, powers[t]);
    dbasis[(size_t)t * m + j] = int_pow(x, powers[t] - 1);
  }
}
void launch_mean_basis(const double* Xs, int m, int D, int nterm, const int* dims, const int* powers, double* basis, double* dbasis, hipStream_t s) {
  hipLaunchKernelGGL(mean_basis_kernel, dim3((m + 255) / 256), dim3(256), 0, s, Xs, m, D, nterm, dims, powers, basis, dbasis);
}

void launch_predict_mean_finish(int nb, int m, int D, int R, int nbasis, const double* basis, const double* coef, const double* dots,
                                const double* LA, double* mean, double* var, long ld, int nterm, const double* dbasis, const int* ddims,
                                const int* dpowers, double* deriv, hipStream_t s) {
  hipLaunchKernelGGL(predict_mean_finish_kernel, dim3((m + 255) / 256, nb), dim3(256), 0, s, nb, m, D, R, nbasis, basis, coef, dots, LA,
                     mean, var, ld, nterm, dbasis, ddims, dpowers, deriv);
}

void launch_trtri_merges(const BatchView& v, hipStream_t s);   // kernels_gemm.hip

void launch_trtri(const BatchView& v, hipStream_t s) {
  hipLaunchKernelGGL(trtri_leaf_kernel, dim3(v.NP / 64, v.nb), dim3(64), 0, s, v, (const double*)v.A);
  launch_trtri_merges(v, s);
}

void launch_extract(const double* src, int NP, int n, double* out, int mode, hipStream_t s) {
  hipLaunchKernelGGL(extract_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, src, NP, n, out, mode);
}

}  // namespace mogp
