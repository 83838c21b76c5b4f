// Device-side 64x64 potf2 executed by ONE wave; shared by the stand-alone potf2 kernel and by the
// MFMA update kernel, whose diagonal-tile workgroup factors the block it has just updated (so the
// latency-bound potf2 runs underneath the rest of that launch instead of as its own launch).
#pragma once
#include "launch.h"

namespace mogp {

typedef double v2d_p __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double readlane_f64(double x, int srclane) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(d) and sqrt(d) by v_rsq_f64 + two coupled Goldschmidt steps (about six dependent FMAs)
// instead of a correctly rounded sqrt followed by a divide (two long dependent sequences) on the
// critical path of every column.  Result error <= ~2 ulp, i.e. backward-stable like LAPACK's.
__device__ __forceinline__ void rsqrt_sqrt(double d, double& rs, double& sq) {
  const double r0 = __builtin_amdgcn_rsq(d);
  double g = d * r0, h = 0.5 * r0;
  double e = __builtin_fma(-g, h, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
  e = __builtin_fma(-g, h, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
  rs = h + h;
  sq = g;
}

// Lpack (per emulator, PACK_STRIDE doubles): [c*64 + q] = L_kk[q][c] (q >= c), [4096 + c] = 1/L_kk[c][c],
// [PACK_INV + b*256 + k*16 + i] = inv(L_bb)[i][k] for the four 16x16 diagonal sub-blocks b (MFMA panel TRSM).
// Written by potf2, read by every panel-TRSM workgroup.
constexpr int PACK_INV = 64 * 64 + 64;
constexpr int PACK_STRIDE = PACK_INV + 4 * 256;


constexpr int POTF2_LDS_DOUBLES = 64 * 65 + 16 * 64;   // block image + 16 finished columns

// blk: LDS image of the 64x64 block (row stride 65), already filled by the caller and visible to
// this wave.  Factors it, writes L (lower, upper zeroed) to A, the packed transposed block + reciprocal
// diagonal to `pack`, and the first failing column (1-based, offset c0) to *info_slot if it is 0.
__device__ __forceinline__ void potf2_wave(double* blk, double* colbuf, double* A, int ld, double* pack, int* info_slot, int c0) {
  const int lane = threadIdx.x & 63;
  double a[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) a[c] = blk[lane * 65 + c];
  __builtin_amdgcn_wave_barrier();
  int fail = 0;
  double myrs = 1.0;
  // 4 block steps of 16 columns: inside a block the pivots / multipliers travel by v_readlane
  // (at most 15 per column); the rank-16 update of all later columns takes its broadcast operands
  // from an 8 KB LDS image of the 16 finished columns as aligned ds_read_b128 pairs.
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
#pragma unroll
    for (int jl = 0; jl < 16; ++jl) {
      const int j = jb * 16 + jl;
      double d = readlane_f64(a[j], j);
      if (!(d > 0.0) || !(d < 1e308)) {   // wave-uniform; catches <= 0, NaN and Inf
        if (fail == 0) fail = j + 1;
        d = 1.0;
      }
      double rs, dj;
      rsqrt_sqrt(d, rs, dj);
      const double l = (lane == j) ? dj : a[j] * rs;
      a[j] = l;
      if (lane == j) myrs = rs;
      colbuf[jl * 64 + lane] = l;
#pragma unroll
      for (int c = j + 1; c < jb * 16 + 16; ++c) a[c] = __builtin_fma(-l, readlane_f64(l, c), a[c]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = jb * 16 + 16; c < 64; c += 2) {
#pragma unroll
      for (int jl = 0; jl < 16; ++jl) {
        const v2d_p lc = *reinterpret_cast<const v2d_p*>(&colbuf[jl * 64 + c]);
        a[c] = __builtin_fma(-a[jb * 16 + jl], lc[0], a[c]);
        a[c + 1] = __builtin_fma(-a[jb * 16 + jl], lc[1], a[c + 1]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    const double x = (c <= lane) ? a[c] : 0.0;      // upper triangle of the block is written as zeros
    blk[lane * 65 + c] = x;
    if (c <= lane) pack[c * 64 + lane] = x;           // column c of L, coalesced across lanes
  }
  pack[4096 + lane] = myrs;
  colbuf[lane] = myrs;
  __builtin_amdgcn_wave_barrier();
  {
    // inverses of the four 16x16 diagonal sub-blocks: lane 16b + j solves L_bb z = e_j (forward substitution,
    // 136 FMAs); the panel TRSM multiplies with them on the matrix cores instead of substituting per row
    const int b = lane >> 4, j = lane & 15;
    const double* Lb = blk + (16 * b) * 65 + 16 * b;
    double zc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) s = __builtin_fma(-Lb[i * 65 + k], zc[k], s);     // z_k = 0 for k < j
      zc[i] = (i < j) ? 0.0 : s * colbuf[16 * b + i];
    }
    double* dst = pack + PACK_INV + b * 256 + j * 16;
#pragma unroll
    for (int i = 0; i < 16; i += 2) *reinterpret_cast<v2d_p*>(dst + i) = (v2d_p){zc[i], zc[i + 1]};
  }
  __builtin_amdgcn_wave_barrier();
  {
    const int half = lane >> 5, part = lane & 31;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) {
      const int r = 2 * q + half;
      v2d_p w;
      w[0] = blk[r * 65 + 2 * part];
      w[1] = blk[r * 65 + 2 * part + 1];
      *reinterpret_cast<v2d_p*>(A + (size_t)r * ld + 2 * part) = w;
    }
  }
  if (lane == 0 && fail != 0 && *info_slot == 0) *info_slot = c0 + fail;
}

// ---------------------------------------------------------------------------------------------
// 64x64 potf2 by a 256-thread workgroup: wave w owns the 16 columns [16w, 16w+16) of all 64 rows
// (lane = row, 16 values per lane instead of 64: ~60 VGPRs, so the routine can live inside the MFMA
// update kernel without costing it occupancy).  Block step b: wave b factors its 16 columns exactly
// like potf2_wave does inside a block (pivot / multipliers by v_readlane), publishes them through
// an 8 KB LDS image, and the waves to its right apply the rank-16 update to their own columns in
// parallel.  Arithmetic (operation order per element) is identical to potf2_wave.
//   lds: POTF2B_LDS_DOUBLES doubles.  All 256 threads must call.
// ---------------------------------------------------------------------------------------------
constexpr int POTF2B_LDS_DOUBLES = 16 * 64 + 4 * 16 * 17 + 64;   // finished columns + diagonal blocks + reciprocal diagonal

__device__ __forceinline__ void potf2_block_dev(double* A, int ld, double* pack, int* info_slot, int c0, double* lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double* colbuf = lds;                    // [16][64]
  double* dblk = lds + 16 * 64;            // [4][16][17]
  double* rdg = dblk + 4 * 16 * 17;        // [64]
  double a[16];
  {
    const v2d_p* src = reinterpret_cast<const v2d_p*>(A + (size_t)lane * ld + 16 * w);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const v2d_p x = src[q];
      a[2 * q] = x[0];
      a[2 * q + 1] = x[1];
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (w == b) {
      int fail = 0;
#pragma unroll
      for (int jl = 0; jl < 16; ++jl) {
        const int j = 16 * b + jl;
        double d = readlane_f64(a[jl], j);
        if (!(d > 0.0) || !(d < 1e308)) {   // wave-uniform; catches <= 0, NaN and Inf
          if (fail == 0) fail = j + 1;
          d = 1.0;
        }
        double rs, dj;
        rsqrt_sqrt(d, rs, dj);
        const double l = (lane == j) ? dj : a[jl] * rs;
        a[jl] = l;
        if (lane == j) rdg[j] = rs;
        colbuf[jl * 64 + lane] = l;
#pragma unroll
        for (int c = jl + 1; c < 16; ++c) a[c] = __builtin_fma(-l, readlane_f64(l, 16 * b + c), a[c]);
      }
      if (lane >= 16 * b && lane < 16 * b + 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) dblk[(b * 16 + (lane - 16 * b)) * 17 + c] = a[c];
      }
      if (lane == 0 && fail != 0 && *info_slot == 0) *info_slot = c0 + fail;
    }
    __syncthreads();
    if (w > b) {
      double mine[16];
#pragma unroll
      for (int jl = 0; jl < 16; ++jl) mine[jl] = colbuf[jl * 64 + lane];
#pragma unroll
      for (int c = 0; c < 16; c += 2) {
#pragma unroll
        for (int jl = 0; jl < 16; ++jl) {
          const v2d_p lc = *reinterpret_cast<const v2d_p*>(&colbuf[jl * 64 + 16 * w + c]);
          a[c] = __builtin_fma(-mine[jl], lc[0], a[c]);
          a[c + 1] = __builtin_fma(-mine[jl], lc[1], a[c + 1]);
        }
      }
    }
    __syncthreads();
  }
  // L (lower, upper zeroed) back to A; packed transposed copy for the panel TRSM
  {
    v2d_p* dst = reinterpret_cast<v2d_p*>(A + (size_t)lane * ld + 16 * w);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = 16 * w + 2 * q;
      v2d_p x;
      x[0] = (c <= lane) ? a[2 * q] : 0.0;
      x[1] = (c + 1 <= lane) ? a[2 * q + 1] : 0.0;
      dst[q] = x;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (16 * w + c <= lane) pack[(16 * w + c) * 64 + lane] = a[c];
  }
  if (w == 0) {
    pack[4096 + lane] = rdg[lane];
    // inverses of the four 16x16 diagonal sub-blocks (see potf2_wave)
    const int b = lane >> 4, j = lane & 15;
    const double* Lb = dblk + b * 16 * 17;
    double zc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) s = __builtin_fma(-Lb[i * 17 + k], zc[k], s);
      zc[i] = (i < j) ? 0.0 : s * rdg[16 * b + i];
    }
    double* dst = pack + PACK_INV + b * 256 + j * 16;
#pragma unroll
    for (int i = 0; i < 16; i += 2) *reinterpret_cast<v2d_p*>(dst + i) = (v2d_p){zc[i], zc[i + 1]};
  }
}

}  // namespace mogp
