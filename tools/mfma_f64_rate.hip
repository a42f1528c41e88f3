// Microbenchmark: issue rate of the f64 matrix / vector FMA pipes on gfx950.
// Used to fix roofline.peak for the ADMM kernel (the guide lists no FP64 peak).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  double a = a0, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// mixed: waves 0-3 MFMA, waves 4-7 VALU FMA in the same workgroup (512 threads)
__global__ __launch_bounds__(512) void k_mixed(double* out, int iters, double a0, double b0) {
  int wave = threadIdx.x >> 6;
  double s = 0;
  if (wave < 4) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters * 8; ++it)   // 8x more FMAs: 16x16x4 = 16 wave-FMAs of work
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(acc[i], a0, b0);
    for (int i = 0; i < 8; ++i) s += acc[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int cus = p.multiProcessorCount; double* out; CK(hipMalloc(&out, sizeof(double) * cus * 8 * 512));
  int iters = 20000;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    int grid = cus * bpc;
    float ms;
    ms = timeit([&] { k_mfma16<1><<<grid, 256>>>(out, iters, 1.0, 1e-9); });
    printf("mfma16x16x4 acc1 blocks/CU %d: %.3f ms  %.2f TFLOP/s  cyc/inst@2.4GHz %.1f\n", bpc, ms, 2048.0 * iters * 1 * grid * 4 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 1.0 * bpc));
    ms = timeit([&] { k_mfma16<4><<<grid, 256>>>(out, iters, 1.0, 1e-9); });
    printf("mfma16x16x4 acc4 blocks/CU %d: %.3f ms  %.2f TFLOP/s  cyc/inst %.1f\n", bpc, ms, 2048.0 * iters * 4 * grid * 4 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 4.0 * bpc));
    ms = timeit([&] { k_mfma4<1><<<grid, 256>>>(out, iters, 1.0, 1e-9); });
    printf("mfma4x4x4_4b acc1 blocks/CU %d: %.3f ms  %.2f TFLOP/s  cyc/inst %.1f\n", bpc, ms, 512.0 * iters * 1 * grid * 4 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 1.0 * bpc));
    ms = timeit([&] { k_mfma4<8><<<grid, 256>>>(out, iters, 1.0, 1e-9); });
    printf("mfma4x4x4_4b acc8 blocks/CU %d: %.3f ms  %.2f TFLOP/s  cyc/inst %.1f\n", bpc, ms, 512.0 * iters * 8 * grid * 4 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 8.0 * bpc));
    ms = timeit([&] { k_fma<8><<<grid, 256>>>(out, iters, 0.999999, 1e-9); });
    printf("v_fma_f64 acc8 blocks/CU %d: %.3f ms  %.2f TFLOP/s  cyc/inst %.1f\n", bpc, ms, 128.0 * iters * 8 * grid * 4 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 8.0 * bpc));
  }
  {
    float ms = timeit([&] { k_mixed<<<cus, 512>>>(out, iters, 0.999999, 1e-9); });
    double fl = (2048.0 * iters * 4 * 4 + 128.0 * iters * 8 * 8 * 4) * cus;
    printf("mixed mfma+valu 512thr: %.3f ms  %.2f TFLOP/s total\n", ms, fl / ms * 1e-9);
  }
  return 0;
}
