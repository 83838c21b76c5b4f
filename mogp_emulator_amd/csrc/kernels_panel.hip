// The 128 x 128 diagonal block of the blocked Cholesky (chol128_dev.h); with the 128-wide panel solve below it
// (trsm128_kernel, kernels_gemm.hip) it replaces the panel share of cusolverDnDpotrf (densegp_gpu.hpp:451-474).
//
// This file is compiled with -mllvm -amdgpu-mfma-vgpr-form=1: the kernel reads MFMA results back with VALU
// instructions after every MFMA, and with accumulators in AGPRs the compiler brackets each MFMA with
// v_accvgpr_write / v_accvgpr_read copies (1204 of them, 156 with the VGPR form).  The panel solve keeps the default
// (measured: 25 us per workgroup in VGPR form against 10 us, the operand loads lose their registers).
#include "launch.h"
#include "chol128_dev.h"

namespace mogp {

__device__ __forceinline__ int slot_to_emu_p(const int* idx, int z) { return idx ? idx[z] : z; }

// ---------------------------------------------------------------------------------------------
// 128 x 128 diagonal block: one workgroup per emulator (chol128_dev.h)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol128_kernel(BatchView v, int c0, int* __restrict__ info, double* __restrict__ Lpack128) {
  __shared__ __attribute__((aligned(16))) double lds[C128_LDS_DOUBLES];
  const int emu = slot_to_emu_p(v.idx, blockIdx.x);
  chol128_dev(v.A + (size_t)emu * v.MS + (size_t)c0 * v.LD + c0, v.LD, Lpack128 + (size_t)emu * PACK128_STRIDE, info + emu, c0, lds);
}

size_t lpack128_doubles_per_emulator() { return PACK128_STRIDE; }

// factor the 128 x 128 diagonal block at c0 and solve the panel rows [c0 + 128, NP) below it
void launch_trsm128(const BatchView& v, int c0, const double* Lpack128, hipStream_t s);   // kernels_gemm.hip

void launch_panel128(const BatchView& v, int c0, int* info, double* Lpack128, hipStream_t s) {
  prof_begin("chol_diag128", s);
  hipLaunchKernelGGL(chol128_kernel, dim3(v.nb), dim3(256), 0, s, v, c0, info, Lpack128);
  prof_end("chol_diag128", s, (double)v.nb * 128.0 * 128.0 * 128.0 / 3.0, (double)v.nb * 2.0 * 8.0 * 128.0 * 128.0);
  launch_trsm128(v, c0, Lpack128, s);
}

}  // namespace mogp
