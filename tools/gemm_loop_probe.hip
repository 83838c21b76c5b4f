// Probe of the GEMM main loops of the one-launch Cholesky's tasks (csrc/gemm_dev.h): time per 16-deep k-step of a 64 x 128 task tile with ONE
// and with TWO workgroups per CU, operands resident in L2 / the infinity cache -- the loop alone, no dependencies, no solves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mogp_emulator_amd/csrc tools/gemm_loop_probe.hip -o tools/gemm_loop_probe.bin
//   tools/gemm_loop_probe.bin [nk = 64] [reps = 40]
// Floor: 32 MFMAs of 64 cycles per wave and step = 0.853 us at 2.4 GHz with one wave per SIMD, 1.707 us with two.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_dev.h"
using namespace mogp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int LD = 2048;

// V = 0: mainloop_pf<64,128,2,2,4> (rounds 3-4); V = 1: mainloop_q<64,128,2,2,2> (round 5, the kernel's); V = 2: mainloop_q with the barrier pinned behind the step's last MFMAs; V = 3: mainloop_pf
// with PD = 2.  (The barrier-in-mid-step form of mainloop_q measured here in round 5 was level with V = 1 and is gone: profiles/r05_loop_probe_ab.txt.)
template <int V, int WGS>
__global__ __launch_bounds__(256, WGS) void loop_kernel(const double* __restrict__ M, int nk, int reps, double* __restrict__ out, int write_tile) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int r0 = 64 * (blockIdx.x % 30 + 2), c0 = 128 * (blockIdx.x % 13);
  v4d acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0., 0., 0., 0.};
  for (int rep = 0; rep < reps; ++rep) {
    const double *Ap = M + (size_t)r0 * LD, *Bp = M + (size_t)c0 * LD;
    if (V == 0) mainloop_pf<64, 128, 2, 2, 4>(Ap, LD, Bp, LD, nk, acc, smem);
    else if (V == 1) mainloop_q<64, 128, 2, 2, 2>(Ap, LD, Bp, LD, nk, acc, smem);
    else if (V == 2) mainloop_q<64, 128, 2, 2, 2, 1>(Ap, LD, Bp, LD, nk, acc, smem);
    else if (V == 4) mainloop_q<64, 128, 2, 2, 2, 2>(Ap, LD, Bp, LD, nk, acc, smem);
    else if (V == 5) mainloop_q<64, 128, 2, 2, 2, 3>(Ap, LD, Bp, LD, nk, acc, smem);
    else mainloop_pf<64, 128, 2, 2, 2>(Ap, LD, Bp, LD, nk, acc, smem);
    __syncthreads();
  }
  if (write_tile) {
    // the tile in the accumulator layout of for_each_acc_w
    for_each_acc_w<2, 2, 4>(acc, [&](int row, int col, double x) { out[((size_t)blockIdx.x * 64 + row) * 128 + col] = x; });
  } else {
    double s = 0.;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678) out[0] = s;
  }
}

template <int V, int WGS>
static double run(const double* dM, int nk, int reps, double* dOut, int grid, size_t lds, int write_tile) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<V, WGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop_kernel<V, WGS>), dim3(grid), dim3(256), lds, 0, dM, nk, reps, dOut, write_tile);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms < best) best = ms;
  }
  return best;
}

constexpr int NV = 6;
template <int WGS>
static double run_v(int v, const double* dM, int nk, int reps, double* dOut, int grid, size_t lds, int write_tile) {
  switch (v) {
    case 0: return run<0, WGS>(dM, nk, reps, dOut, grid, lds, write_tile);
    case 1: return run<1, WGS>(dM, nk, reps, dOut, grid, lds, write_tile);
    case 2: return run<2, WGS>(dM, nk, reps, dOut, grid, lds, write_tile);
    case 4: return run<4, WGS>(dM, nk, reps, dOut, grid, lds, write_tile);
    case 5: return run<5, WGS>(dM, nk, reps, dOut, grid, lds, write_tile);
    default: return run<3, WGS>(dM, nk, reps, dOut, grid, lds, write_tile);
  }
}

int main(int argc, char** argv) {
  const int nk = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 40;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  std::vector<double> hM((size_t)LD * LD);
  srand(3);
  for (auto& x : hM) x = rand() / (double)RAND_MAX - 0.5;
  double *dM, *dOut;
  CK(hipMalloc(&dM, hM.size() * 8)); CK(hipMalloc(&dOut, (size_t)2 * ncu * 64 * 128 * 8));
  CK(hipMemcpy(dM, hM.data(), hM.size() * 8, hipMemcpyHostToDevice));
  const size_t lds_pf = (size_t)WCfg<64, 128, 2, 2>::SMEM_DOUBLES * 8, lds_q = (size_t)QCfg<64, 128>::SMEM_DOUBLES * 8;
  const size_t solo = 88 * 1024;        // more than half of the CU's LDS: one workgroup per CU
  // correctness: every variant against a host product of the same tile (one pass)
  {
    std::vector<double> ref((size_t)8 * 64 * 128), got(ref.size());
    for (int b = 0; b < 8; ++b) {
      const int r0 = 64 * (b % 30 + 2), c0 = 128 * (b % 13);
      for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 128; ++j) {
          double s = 0.;
          for (int k = 0; k < nk * 16; ++k) s += hM[(size_t)(r0 + i) * LD + k] * hM[(size_t)(c0 + j) * LD + k];
          ref[((size_t)b * 64 + i) * 128 + j] = s;
        }
    }
    for (int v = 0; v < NV; ++v) {
      run_v<2>(v, dM, nk, 1, dOut, 8, v == 0 || v == 3 ? lds_pf : lds_q, 1);
      CK(hipMemcpy(got.data(), dOut, got.size() * 8, hipMemcpyDeviceToHost));
      double err = 0., mx = 0.;
      for (size_t e = 0; e < ref.size(); ++e) { err = std::fmax(err, std::fabs(got[e] - ref[e])); mx = std::fmax(mx, std::fabs(ref[e])); }
      printf("variant %d: max |tile - host product| = %.3e (max |entry| %.3e)  %s\n", v, err, mx, err < 1e-11 * mx * nk ? "OK" : "WRONG");
    }
  }
  const double steps = (double)nk * reps;
  for (int wgs = 1; wgs <= 2; ++wgs) {
    const int grid = ncu * wgs;
    double ms[NV];
    for (int v = 0; v < NV; ++v) {
      const size_t lds = wgs == 1 ? solo : (v == 0 || v == 3 ? lds_pf : lds_q);
      ms[v] = wgs == 1 ? run_v<1>(v, dM, nk, reps, dOut, grid, lds, 0) : run_v<2>(v, dM, nk, reps, dOut, grid, lds, 0);
    }
    const char* names[NV] = {"mainloop_pf<..,4>", "mainloop_q<..,2> ", "mainloop_q<..,2,PIN>", "mainloop_pf<..,2>", "mainloop_q<..,2,PIN+interleave>", "mainloop_q<..,2,interleave, next fragments early>"};
    for (int v = 0; v < NV; ++v) {
      const double us = ms[v] * 1e3 / steps;
      const double tf = (double)grid * steps * 64. * 128. * 16. * 2. / (ms[v] * 1e-3) * 1e-12;
      printf("%d workgroup(s) per CU, %s: %.3f us per k-step, %.1f TFLOP/s (%.2f of 78.6)\n", wgs, names[v], us, tf, tf / 78.6);
    }
  }
  return 0;
}
