// Probe: (1) empirical lane/register -> (row,col) map of v_mfma_f64_16x16x4_f64 C/D,
//        (2) A/B operand map, (3) dependent/independent issue rate -> fp64 MFMA peak,
//        (4) v_fma_f64 VALU peak, (5) device-to-device copy bandwidth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void layout_kernel(double* out_row, double* out_col, double* out_chk) {
  int l = threadIdx.x;
  // A[i][k] = (k==0) ? i : 0  assuming A operand lane map i=l&15,k=l>>4 ; B[k][j] = (k==0)?1:0
  double a = ((l >> 4) == 0) ? (double)(l & 15) : 0.0;
  double b = ((l >> 4) == 0) ? 1.0 : 0.0;
  v4d c = {0, 0, 0, 0};
  v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);   // D[i][j] = i
  for (int r = 0; r < 4; ++r) out_row[l * 4 + r] = d[r];
  a = ((l >> 4) == 0) ? 1.0 : 0.0;
  b = ((l >> 4) == 0) ? (double)(l & 15) : 0.0;
  d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);       // D[i][j] = j
  for (int r = 0; r < 4; ++r) out_col[l * 4 + r] = d[r];
  // check k mapping: A[i][k] = 10^k-ish weights, B[k][j] = (k+1): D = sum_k A[i][k]*B[k][j]
  a = (double)(1 + (l >> 4)) * 100.0 + (l & 15);       // A[i][k] = 100(k+1) + i
  b = (double)(1 + (l >> 4)) * 0.001 * (1 + (l & 15)); // B[k][j] = 0.001 (k+1)(j+1)
  d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out_chk[l * 4 + r] = d[r];
}

template <int NACC>
__global__ void mfma_rate_kernel(double* out, int iters) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void fma_rate_kernel(double* out, int iters) {
  double x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
  double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void copy_kernel(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz arch %s\n", p.name, p.multiProcessorCount, p.clockRate, p.gcnArchName);
  double *d_row, *d_col, *d_chk; CK(hipMalloc(&d_row, 256 * 8)); CK(hipMalloc(&d_col, 256 * 8)); CK(hipMalloc(&d_chk, 256 * 8));
  layout_kernel<<<1, 64>>>(d_row, d_col, d_chk); CK(hipDeviceSynchronize());
  std::vector<double> row(256), col(256), chk(256);
  CK(hipMemcpy(row.data(), d_row, 256 * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(col.data(), d_col, 256 * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(chk.data(), d_chk, 256 * 8, hipMemcpyDeviceToHost));
  int ok_guide = 1, ok_f32 = 1, ok_chk = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int R = (int)row[l * 4 + r], C = (int)col[l * 4 + r];
    if (R != (l >> 4) + 4 * r || C != (l & 15)) ok_guide = 0;
    if (R != (l >> 4) * 4 + r || C != (l & 15)) ok_f32 = 0;
    double expect = 0; for (int k = 0; k < 4; ++k) expect += (100.0 * (k + 1) + R) * 0.001 * (k + 1) * (C + 1);
    if (fabs(chk[l * 4 + r] - expect) > 1e-9) ok_chk = 0;
  }
  printf("LAYOUT f64 C/D: row=(lane>>4)+4*reg col=lane&15 : %s ; f32-style row=(lane>>4)*4+reg : %s ; A/B k-map check: %s\n",
         ok_guide ? "YES" : "no", ok_f32 ? "YES" : "no", ok_chk ? "OK" : "MISMATCH");
  printf("lane0 rows: %g %g %g %g ; lane16 rows: %g %g %g %g ; lane17 col %g\n", row[0], row[1], row[2], row[3],
         row[64], row[65], row[66], row[67], col[17 * 4]);

  double* d_out; CK(hipMalloc(&d_out, 4096 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return (double)ms; };
  int iters = 20000;
  for (int wpb : {256, 512}) {
    int blocks = p.multiProcessorCount * 2;
    double ms1 = time_it([&] { mfma_rate_kernel<1><<<blocks, wpb>>>(d_out, iters); });
    double ms4 = time_it([&] { mfma_rate_kernel<4><<<blocks, wpb>>>(d_out, iters / 4); });
    double ms8 = time_it([&] { mfma_rate_kernel<8><<<blocks, wpb>>>(d_out, iters / 8); });
    double waves = blocks * (wpb / 64.0);
    double fl = waves * iters * 2048.0;
    printf("MFMA f64 16x16x4: block %d x %d blocks: dep-chain %.1f TF, 4 acc %.1f TF, 8 acc %.1f TF\n", wpb, blocks,
           fl / ms1 * 1e-9, fl / ms4 * 1e-9, fl / ms8 * 1e-9);
  }
  {
    int blocks = p.multiProcessorCount * 4, wpb = 256, it2 = 20000;
    double ms = time_it([&] { fma_rate_kernel<<<blocks, wpb>>>(d_out, it2); });
    printf("VALU v_fma_f64: %.1f TF\n", (double)blocks * wpb * it2 * 16 * 2 / ms * 1e-9);
  }
  {
    size_t n = (size_t)1 << 27;  // 2 GiB of double2
    double2 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 1, n * 16));
    double ms = time_it([&] { copy_kernel<<<p.multiProcessorCount * 8, 256>>>(a, b, n); });
    printf("copy 2x%.1f GiB: %.2f TB/s (read+write)\n", n * 16.0 / (1 << 30), 2.0 * n * 16 / ms * 1e-9);
  }
  return 0;
}
