// fp64 MFMA issue / latency probe: cycles per v_mfma_f64_16x16x4_f64 for one wave as a function of the number of
// independent accumulators and of the number of waves sharing the CU (is the fp64 matrix pipe per SIMD or per CU?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(double* out, unsigned long long* cyc, int iters) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int threads, int blocks) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double per_wave_ns = ms * 1e6 / ((double)iters * NACC);
  printf("NACC=%d waves/WG=%d blocks=%d : %.1f counter ticks per MFMA per wave, %.1f ns per MFMA per wave (wall), chip %.1f TFLOP/s\n", NACC, threads / 64, blocks,
         (double)h / (iters * NACC), per_wave_ns, 2048.0 * iters * NACC * (threads / 64) * blocks / (ms * 1e-3) * 1e-12);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {64, 128, 256, 512}) {
    run<1>(threads, 1); run<2>(threads, 1); run<4>(threads, 1); run<8>(threads, 1);
  }
  run<4>(256, 256); run<4>(512, 256); run<4>(1024, 256);
  return 0;
}
