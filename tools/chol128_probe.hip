// Stand-alone probe of the 128 x 128 diagonal-block kernel (csrc/chol128_dev.h): factors NB random SPD blocks, checks
// L against a host Cholesky, reports the kernel time (HIP events) and in-kernel cycle stamps per phase.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DC128_PROFILE -I mogp_emulator_amd/csrc tools/chol128_probe.hip -o tools/chol128_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "chol128_dev.h"
using namespace mogp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe_kernel(double* A, int ld, size_t stride, double* pk, int* info, unsigned long long* stamps) {
  __shared__ __attribute__((aligned(16))) double lds[C128_LDS_DOUBLES];
#ifdef C128_PROFILE
  if (threadIdx.x == 0 && blockIdx.x == 0) c128_stamps = stamps;
  c128_stamps = stamps;
#endif
  chol128_dev(A + blockIdx.x * stride, ld, pk + (size_t)blockIdx.x * PACK128_STRIDE, info + blockIdx.x, 0, lds);
}

int main(int argc, char** argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 64, ld = 2048;
  const double shift = argc > 2 ? atof(argv[2]) : 8.0;      // small shift + low-rank G: ill-conditioned blocks
  const int rank = argc > 3 ? atoi(argv[3]) : 128;
  const size_t stride = (size_t)128 * ld;
  std::vector<double> hA(NB * stride), hL(NB * stride);
  srand(1);
  for (int b = 0; b < NB; ++b) {
    // SPD: G G^T + 128 I from a random G, lower triangle stored (upper left as garbage)
    std::vector<double> G(128 * 128);
    for (auto& x : G) x = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 128; ++i)
      for (int j = 0; j < 128; ++j) {
        double s = (i == j) ? shift : 0.0;
        for (int k = 0; k < rank; ++k) s += G[i * 128 + k] * G[j * 128 + k];
        hA[b * stride + (size_t)i * ld + j] = (j <= i) ? s : 1e30;
      }
  }
  double *dA, *dpk; int* dinfo; unsigned long long* dst;
  CK(hipMalloc(&dA, hA.size() * 8)); CK(hipMalloc(&dpk, (size_t)NB * PACK128_STRIDE * 8)); CK(hipMalloc(&dinfo, NB * 4));
  CK(hipMalloc(&dst, (2048 + NB * 4 * 64) * 8));
  CK(hipMemset(dinfo, 0, NB * 4)); CK(hipMemset(dst, 0, (2048 + NB * 4 * 64) * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 6; ++it) {
    CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe_kernel, dim3(NB), dim3(256), 0, 0, dA, ld, stride, dpk, dinfo, dst);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 0 && ms < best) best = ms;
  }
  CK(hipMemcpy(hL.data(), dA, hA.size() * 8, hipMemcpyDeviceToHost));
  std::vector<int> info(NB); CK(hipMemcpy(info.data(), dinfo, NB * 4, hipMemcpyDeviceToHost));
  std::vector<double> pk((size_t)NB * PACK128_STRIDE); CK(hipMemcpy(pk.data(), dpk, pk.size() * 8, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> st(2048 + NB * 4 * 64); CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
  // host check
  double maxerr = 0., maxinv = 0., maxlt = 0., res_dev = 0., res_host = 0.;
  for (int b = 0; b < NB; ++b) {
    std::vector<double> L(128 * 128, 0.);
    for (int j = 0; j < 128; ++j) {
      double d = hA[b * stride + (size_t)j * ld + j];
      for (int k = 0; k < j; ++k) d -= L[j * 128 + k] * L[j * 128 + k];
      d = std::sqrt(d);
      L[j * 128 + j] = d;
      for (int i = j + 1; i < 128; ++i) {
        double s = hA[b * stride + (size_t)i * ld + j];
        for (int k = 0; k < j; ++k) s -= L[i * 128 + k] * L[j * 128 + k];
        L[i * 128 + j] = s / d;
      }
    }
    for (int i = 0; i < 128; ++i)
      for (int j = 0; j < 128; ++j) {
        if (j < 64 || i >= 64)      // the upper-right 64 x 64 tile is not part of the post-condition
          maxerr = std::fmax(maxerr, std::fabs(hL[b * stride + (size_t)i * ld + j] - L[i * 128 + j]));
        if (j <= i) maxlt = std::fmax(maxlt, std::fabs(pk[(size_t)b * PACK128_STRIDE + PACK128_LT + j * 128 + i] - L[i * 128 + j]));
      }
    // backward error: max |L L^T - A| / max |A| for the device factor and for the host factor
    {
      double amax = 0., rd = 0., rh = 0.;
      for (int i = 0; i < 128; ++i)
        for (int j = 0; j <= i; ++j) {
          double sd = 0., sh = 0.;
          for (int k = 0; k <= j; ++k) {
            sd += hL[b * stride + (size_t)i * ld + k] * hL[b * stride + (size_t)j * ld + k];
            sh += L[i * 128 + k] * L[j * 128 + k];
          }
          const double a = hA[b * stride + (size_t)i * ld + j];
          amax = std::fmax(amax, std::fabs(a));
          rd = std::fmax(rd, std::fabs(sd - a));
          rh = std::fmax(rh, std::fabs(sh - a));
        }
      res_dev = std::fmax(res_dev, rd / amax);
      res_host = std::fmax(res_host, rh / amax);
    }
    // inv check: L_bb * inv = I
    for (int bb = 0; bb < 8; ++bb)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double s = 0.;
          for (int k = 0; k < 16; ++k) s += L[(16 * bb + i) * 128 + 16 * bb + k] * pk[(size_t)b * PACK128_STRIDE + PACK128_INV + bb * 256 + j * 16 + k];
          maxinv = std::fmax(maxinv, std::fabs(s - (i == j ? 1.0 : 0.0)));
        }
  }
  printf("NB=%d kernel %.1f us  max|L - Lref| %.3e  max|LT - Lref| %.3e  max|L inv - I| %.3e  info[0]=%d\n", NB, best * 1e3, maxerr, maxlt, maxinv, info[0]);
  printf("  backward error max|L L^T - A| / max|A|: device %.3e, host Cholesky %.3e\n", res_dev, res_host);
  const char* names[19] = {"start", "loaded", "b0 cols", "b0 syrk", "b1 cols", "b1 syrk", "b2 cols", "b2 syrk", "b3 cols", "b3 syrk", "b4 cols", "b4 syrk",
                           "b5 cols", "b5 syrk", "b6 cols", "b6 syrk", "b7 cols", "inverses", "stored"};
  for (int i = 1; i < 19; ++i)
    if (st[i] && st[i - 1]) printf("  %-9s %8llu cycles\n", names[i], st[i] - st[i - 1]);
  printf("  total     %8llu cycles (readcyclecounter units)\n", st[18] - st[0]);
  // per-wave s_memrealtime stamps (100 MHz): block 0, waves 0..3: [start steps, end steps, end put, after barrier] per block step
  for (int w = 0; w < 4; ++w) {
    const unsigned long long* q = st.data() + 2048 + (0 * 4 + w) * 64;
    printf("  wave %d (x10 ns): ", w);
    for (int b = 0; b < 8; ++b)
      if (q[4 * b] && q[4 * b + 1]) printf("b%d: steps %llu put %llu bar %llu syrk %llu | ", b, q[4 * b + 1] - q[4 * b], q[4 * b + 2] - q[4 * b + 1],
                                           q[4 * b + 3] > q[4 * b + 2] ? q[4 * b + 3] - q[4 * b + 2] : 0ULL, (b < 7 && q[4 * b + 4] > q[4 * b + 3]) ? q[4 * b + 4] - q[4 * b + 3] : 0ULL);
    printf("\n");
    printf("  wave %d update phase (x10 ns): ", w);
    for (int b = 0; b < 7; ++b)
      if (q[32 + 4 * b] && q[4 * b + 3]) printf("b%d: copy-out %llu mfma %llu publish+barrier %llu read-D %llu | ", b, q[32 + 4 * b] - q[4 * b + 3], q[33 + 4 * b] - q[32 + 4 * b],
                                                q[34 + 4 * b] - q[33 + 4 * b], q[4 * b + 4] > q[34 + 4 * b] ? q[4 * b + 4] - q[34 + 4 * b] : 0ULL);
    printf("\n");
  }
  return 0;
}
