// 128-wide panel solve below a factored 128 x 128 diagonal block, on the matrix cores.
#pragma once
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
// (L_ba or inv(L_bb), element [lane&15][g+4r]) is fetched from the pack written by chol128_dev with the same k mapping.
// The rank-64 update of the second half of the block column with the first is part of the substitution, so the panel
// is read and written once, as full 1 KB rows through a wave-private LDS stage (the MFMA C/D layout would otherwise make
// a lane touch 8-byte pieces 32 bytes apart, overflowing the L1 with partially used lines).
constexpr int TRSM128_STAGE = 16 * 130;      // wave-private slab of 16 rows x 128 columns, row stride 130 (conflict free)

__device__ __forceinline__ void trsm128_dev(const BatchView& v, int c0, int r0, const double* __restrict__ pk, int emu, int rowblock,
                                            double* stage) {
  const int ld = v.LD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, i = lane & 15;
  double* slab = v.A + (size_t)emu * v.MS + (size_t)(r0 + rowblock * 64 + wave * 16) * ld + c0;   // 16 rows x 128 columns
  stage += wave * TRSM128_STAGE;
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
  const double* LT = pk + PACK128_LT;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
#pragma unroll
    for (int a = 0; a < b; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r)      // A operand: L[16b + i][16a + g + 4r]
        T[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-LT[(16 * a + g + 4 * r) * 128 + 16 * b + i], X[a][r], T[b], 0, 0, 0);
    const double* inv = pk + PACK128_INV + b * 256;
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
