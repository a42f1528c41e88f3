// Probe of v_mfma_f64_16x16x4 on gfx950: operand / result lane layout and the issue interval of one wave (cycles between
// back-to-back independent instructions, s_memtime), next to v_mfma_f64_4x4x4.  hipcc --offload-arch=gfx950 -O3 -o /tmp/p16 tools/mfma_f64_16x16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_layout(const double* A, const double* B, double* D) {   // A: 64 lane values, B: 64 lane values; D: 64 x 4
  const int l = threadIdx.x;
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l], B[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}
template <int NACC>
__global__ void k_rate16(unsigned long long* out, double* sink, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[threadIdx.x] = s;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
template <int NACC>
__global__ void k_rate4(unsigned long long* out, double* sink, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[threadIdx.x] = s;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
int main() {
  double hA[64], hB[64], hD[256];
  // assumed layout: A[i][k] at lane 16 k + i, B[k][j] at lane 16 k + j, D[4 (l / 16) + r][l % 16] in register r of lane l
  double Am[16][4], Bm[4][16];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) Am[i][k] = 1 + i + 0.1 * k;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) Bm[k][j] = 2 + 0.01 * j + 3 * k;
  for (int l = 0; l < 64; ++l) { hA[l] = Am[l % 16][l / 16]; hB[l] = Bm[l / 16][l % 16]; }
  double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
  CK(hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD); CK(hipDeviceSynchronize());
  CK(hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l / 16) + r, j = l % 16;
    double ref = 0; for (int k = 0; k < 4; ++k) ref += Am[i][k] * Bm[k][j];
    worst = fmax(worst, fabs(ref - hD[l * 4 + r]));
  }
  printf("layout A[i][k]@16k+i, B[k][j]@16k+j, D[4(l/16)+r][l%%16]: max error %.3e %s\n", worst, worst < 1e-9 ? "(confirmed)" : "(WRONG)");
  if (worst >= 1e-9) {          // find (i, j) of every (lane, register) under the assumed A / B layouts
    for (int l = 0; l < 64; l += (l < 4 ? 1 : 15)) for (int r = 0; r < 4; ++r) {
      int fi = -1, fj = -1;
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double ref = 0; for (int k = 0; k < 4; ++k) ref += Am[i][k] * Bm[k][j]; if (fabs(ref - hD[l * 4 + r]) < 1e-9) { fi = i; fj = j; } }
      printf("  lane %2d reg %d -> D[%d][%d]\n", l, r, fi, fj);
    }
  }
  unsigned long long* dout; double* sink; CK(hipMalloc(&dout, 8)); CK(hipMalloc(&sink, 512));
  unsigned long long c;
  const int iters = 2000;
#define RUN(K, N) do { hipLaunchKernelGGL((K<N>), dim3(1), dim3(64), 0, 0, dout, sink, iters); CK(hipDeviceSynchronize()); hipLaunchKernelGGL((K<N>), dim3(1), dim3(64), 0, 0, dout, sink, iters); CK(hipDeviceSynchronize()); \
    CK(hipMemcpy(&c, dout, 8, hipMemcpyDeviceToHost)); printf("%-8s %d independent accumulators: %.1f clocks (s_memtime) per instruction\n", #K, N, (double)c / (iters * (double)N)); } while (0)
  RUN(k_rate16, 1); RUN(k_rate16, 2); RUN(k_rate16, 6);
  RUN(k_rate4, 1); RUN(k_rate4, 4); RUN(k_rate4, 8);
  return 0;
}
