// Panel TRSM on the matrix cores (device function shared by the stand-alone kernel and the role-fused step kernel).
#pragma once
#include "launch.h"
#include "potf2_dev.h"

namespace mogp {

typedef double v4d_t __attribute__((ext_vector_type(4)));

// stage: optional wave-private LDS slab of TRSM_STAGE doubles per wave.  The MFMA C/D layout makes a lane touch eight
// 8-byte pieces of its row, 32 bytes apart -- with four waves per workgroup that working set overflows the L1 and every
// 128-byte line is fetched several times.  Staged, the slab is read and written as full 512-byte rows (16-byte pieces,
// two rows per instruction) and transposed into the MFMA layout through LDS (row stride 66 doubles: conflict free).
constexpr int TRSM_STAGE = 16 * 66;

__device__ __forceinline__ void trsm_mfma_pk(const BatchView& v, int c0, int r0, const double* __restrict__ pk, int emu, int rowblock,
                                             double* stage = nullptr) {
  const int ld = v.LD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, i = lane & 15;
  const int row = r0 + rowblock * 64 + wave * 16 + i;
  double* arow = v.A + (size_t)emu * v.MS + (size_t)row * ld + c0;
  double* slab = v.A + (size_t)emu * v.MS + (size_t)(r0 + rowblock * 64 + wave * 16) * ld + c0;   // 16 rows x 64 columns
  if (stage) {
    stage += wave * TRSM_STAGE;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = lane + 64 * q, rl = c >> 5, ch = c & 31;
      *reinterpret_cast<v2d_p*>(stage + rl * 66 + 2 * ch) = *reinterpret_cast<const v2d_p*>(slab + (size_t)rl * ld + 2 * ch);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // A operands: Lneg[b][a][r] = -L[16b + i][16a + g + 4r]  (a < b),  Inv[b][r] = inv(L_bb)[i][g + 4r]
  double Lneg[6][4], Inv[4][4];
#pragma unroll
  for (int b = 1; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < b; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) Lneg[b * (b - 1) / 2 + a][r] = -pk[(16 * a + g + 4 * r) * 64 + 16 * b + i];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) Inv[b][r] = pk[PACK_INV + b * 256 + (g + 4 * r) * 16 + i];
  v4d_t T[4], X[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) T[b][r] = stage ? stage[i * 66 + 16 * b + g + 4 * r] : arow[16 * b + g + 4 * r];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int a = 0; a < b; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lneg[b * (b - 1) / 2 + a][r], X[a][r], T[b], 0, 0, 0);
    X[b] = (v4d_t){0., 0., 0., 0.};
#pragma unroll
    for (int r = 0; r < 4; ++r) X[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(Inv[b][r], T[b][r], X[b], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (stage) stage[i * 66 + 16 * b + g + 4 * r] = X[b][r];
      else arow[16 * b + g + 4 * r] = X[b][r];
    }
  }
  if (stage) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = lane + 64 * q, rl = c >> 5, ch = c & 31;
      *reinterpret_cast<v2d_p*>(slab + (size_t)rl * ld + 2 * ch) = *reinterpret_cast<const v2d_p*>(stage + rl * 66 + 2 * ch);
    }
  }
}


__device__ __forceinline__ void trsm_mfma_dev(const BatchView& v, int c0, int r0, const double* __restrict__ Lpack, int emu, int rowblock,
                                              double* stage = nullptr) {
  trsm_mfma_pk(v, c0, r0, Lpack + (size_t)emu * PACK_STRIDE, emu, rowblock, stage);
}

// ---------------------------------------------------------------------------------------------
// 128-wide panel solve X L^T = B for a factored 128 x 128 diagonal block L = [L11 0; L21 L22]: the same block forward
// substitution over EIGHT 16-column blocks, so the rank-64 update of the second half with the first is part of the
// substitution (no separate 64-wide update launch, the panel is read and written once instead of 7/4 times).
//   pack128: [0, PACK_STRIDE) pack of L11, [PACK_STRIDE, 2 PACK_STRIDE) pack of L22, then [c*64 + q] = L21[q][c]
// ---------------------------------------------------------------------------------------------
constexpr int PACK128_STRIDE = 2 * PACK_STRIDE + 64 * 64;

__device__ __forceinline__ double pack128_L(const double* __restrict__ pk, int br, int bc, int i, int k) {
  // L[16 br + i][16 bc + k] of the 128 x 128 block, br > bc
  if (br < 4) return pk[(16 * bc + k) * 64 + 16 * br + i];
  if (bc >= 4) return pk[PACK_STRIDE + (16 * (bc - 4) + k) * 64 + 16 * (br - 4) + i];
  return pk[2 * PACK_STRIDE + (16 * bc + k) * 64 + 16 * (br - 4) + i];
}

constexpr int TRSM128_STAGE = 16 * 130;      // wave-private slab of 16 rows x 128 columns, row stride 130 (conflict free)

__device__ __forceinline__ void trsm128_dev(const BatchView& v, int c0, int r0, const double* __restrict__ pk, int emu, int rowblock,
                                            double* stage) {
  const int ld = v.LD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, i = lane & 15;
  double* slab = v.A + (size_t)emu * v.MS + (size_t)(r0 + rowblock * 64 + wave * 16) * ld + c0;   // 16 rows x 128 columns
  stage += wave * TRSM128_STAGE;
  // full 1 KB rows in, 16 bytes per lane (see trsm_mfma_pk)
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int c = lane + 64 * q, rl = c >> 6, ch = c & 63;
    *reinterpret_cast<v2d_p*>(stage + rl * 130 + 2 * ch) = *reinterpret_cast<const v2d_p*>(slab + (size_t)rl * ld + 2 * ch);
  }
  __builtin_amdgcn_wave_barrier();
  v4d_t T[8], X[8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) T[b][r] = stage[i * 130 + 16 * b + g + 4 * r];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
#pragma unroll
    for (int a = 0; a < b; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pack128_L(pk, b, a, i, g + 4 * r), X[a][r], T[b], 0, 0, 0);
    const double* inv = pk + (b < 4 ? 0 : PACK_STRIDE) + PACK_INV + (b & 3) * 256;
    X[b] = (v4d_t){0., 0., 0., 0.};
#pragma unroll
    for (int r = 0; r < 4; ++r) X[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(inv[(g + 4 * r) * 16 + i], T[b][r], X[b], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) stage[i * 130 + 16 * b + g + 4 * r] = X[b][r];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int c = lane + 64 * q, rl = c >> 6, ch = c & 63;
    *reinterpret_cast<v2d_p*>(slab + (size_t)rl * ld + 2 * ch) = *reinterpret_cast<const v2d_p*>(stage + rl * 130 + 2 * ch);
  }
}

}  // namespace mogp
