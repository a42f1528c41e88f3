// Cycles per block reduction of three values over 256 threads: the DPP tree of tz_block_reduce3 against LDS atomics (ds_max_f64 / ds_add_f64
// into one slot per wave, then a fixed-order combine): tools/bin/reduce_probe   (hipcc --offload-arch=gfx950 -O3 -I tzddpc_amd/csrc tools/reduce_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "tz_kernels.hip.h"

__device__ inline void lds_max(double* a, double v) { asm volatile("ds_max_f64 %0, %1" :: "v"((unsigned)(size_t)a), "v"(v) : "memory"); }
__device__ inline void lds_add(double* a, double v) { asm volatile("ds_add_f64 %0, %1" :: "v"((unsigned)(size_t)a), "v"(v) : "memory"); }

template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(double* out, unsigned long long* cyc, int reps) {
  __shared__ double red[64];
  __shared__ double slot[2][16];
  const int t = threadIdx.x, w = t >> 6;
  double a = 1.0 + t * 1e-3, b = 2.0 - t * 1e-3, c = t * 1e-6;
  int par = 0;
  if (t < 32) ((double*)slot)[t] = (t % 16 < 8) ? -1e300 : 0.0;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {
      tz_block_reduce3<RED_MAX, RED_MAX, RED_SUM>(a, b, c, red, par);
    } else {
      double* s = slot[par];
      lds_max(s + w, a); lds_max(s + 4 + w, b); lds_add(s + 8 + w, c);
      __syncthreads();
      const double a0 = s[0], a1 = s[1], a2 = s[2], a3 = s[3], b0 = s[4], b1 = s[5], b2 = s[6], b3 = s[7], c0 = s[8], c1 = s[9], c2 = s[10], c3 = s[11];
      a = fmax(fmax(a0, a1), fmax(a2, a3)); b = fmax(fmax(b0, b1), fmax(b2, b3)); c = (c0 + c1) + (c2 + c3);
      par ^= 1;
      if (t < 12) slot[par][t] = (t < 8) ? -1e300 : 0.0;     // the other buffer: last read before the barrier above
    }
    a = a * 0.999 + t * 1e-9; b = b * 0.999 + 1e-9 * t; c = c * 1e-3 + t * 1e-9;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + t] = a + b + c;
}

int main() {
  const int B = 1024, reps = 2000;
  double* out; unsigned long long* cyc;
  hipMalloc(&out, B * 256 * 8); hipMalloc(&cyc, B * 8);
  for (int mode = 0; mode < 2; ++mode) {
    for (int it = 0; it < 2; ++it) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(B), dim3(256), 0, 0, out, cyc, reps);
      else hipLaunchKernelGGL(probe<1>, dim3(B), dim3(256), 0, 0, out, cyc, reps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<double> h(4); hipMemcpy(h.data(), out, 32, hipMemcpyDeviceToHost);
      printf("mode %d (%s): %.3f ms for %d reductions of 3 values x %d workgroups (4 per CU) = %.3f us per reduction; out %.17g\n", mode, mode ? "LDS atomics" : "DPP tree", ms, reps, B, ms * 1e3 / reps, h[0]);
    }
  }
  return 0;
}
