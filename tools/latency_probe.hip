// Latency probe for the primitives on the tz_ipm critical path (one workgroup of 256 threads, s_memtime deltas).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
__global__ void probe(double* out, unsigned long long* t, double seed) {
  __shared__ double lds[1024];
  __shared__ int chain[1024];
  int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 256) { lds[i] = seed + i * 1e-9; chain[i] = (i * 37 + 11) & 1023; }
  __syncthreads();
  unsigned long long t0, t1;
  double v = seed; int idx = tid;
  // (a) dependent LDS read chain (index chasing)
  t0 = now();
  for (int i = 0; i < N; ++i) idx = chain[idx];
  t1 = now(); if (tid == 0) t[0] = (t1 - t0) / N; out[tid] = idx;
  // (b) dependent f64 FMA chain
  t0 = now();
  for (int i = 0; i < N; ++i) v = __builtin_fma(v, 0.999999, 1e-9);
  t1 = now(); if (tid == 0) t[1] = (t1 - t0) / N; out[256 + tid] = v;
  // (c) __syncthreads back to back
  t0 = now();
  for (int i = 0; i < N; ++i) __syncthreads();
  t1 = now(); if (tid == 0) t[2] = (t1 - t0) / N;
  // (d) rsq + 2 Newton (as in tz_sqrt_rsqrt), dependent
  double d = 2.0 + seed;
  t0 = now();
  for (int i = 0; i < N; ++i) {
    double y = __builtin_amdgcn_rsq(d); double g = d * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5); g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5); g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double e = __builtin_fma(-g, g, d); g = __builtin_fma(e, h, g);
    double inv = h + h; double e2 = __builtin_fma(-g, inv, 1.0); inv = __builtin_fma(e2, inv, inv);
    d = g + inv + 1.0;
  }
  t1 = now(); if (tid == 0) t[3] = (t1 - t0) / N; out[512 + tid] = d;
  // (e) dependent MFMA 4x4x4 chain
  double acc = 0.0;
  t0 = now();
  for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1e-3, acc, 0, 0, 0);
  t1 = now(); if (tid == 0) t[4] = (t1 - t0) / N; out[768 + tid] = acc;
  // (f) LDS write -> barrier -> LDS read of another thread's value -> fma (one "hop")
  t0 = now();
  for (int i = 0; i < N; ++i) { lds[tid] = v; __syncthreads(); v = __builtin_fma(lds[(tid + 64) & 255], 0.5, v * 0.5); __syncthreads(); }
  t1 = now(); if (tid == 0) t[5] = (t1 - t0) / N; out[1024 + tid] = v;
  // (g) same hop inside one wave: write -> wave fence -> read neighbour
  t0 = now();
  if (tid < 64) for (int i = 0; i < N; ++i) {
    lds[tid] = v; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    v = __builtin_fma(lds[(tid + 5) & 63], 0.5, v * 0.5); }
  t1 = now(); if (tid == 0) t[6] = (t1 - t0) / N; out[1280 + tid] = v;
  // (h) f64 division chain
  t0 = now();
  for (int i = 0; i < N; ++i) v = 1.0 / (v + 1.5);
  t1 = now(); if (tid == 0) t[7] = (t1 - t0) / N; out[1536 + tid] = v;
  // (i) DPP quad broadcast chain (two 32-bit movs)
  t0 = now();
  for (int i = 0; i < N; ++i) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x55, 0xf, 0xf, false);
    unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x55, 0xf, 0xf, false);
    v = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo) + 1e-9; }
  t1 = now(); if (tid == 0) t[8] = (t1 - t0) / N; out[1792 + tid] = v;
  // (j) dependent global load chain (L2 hit)
  const double* gp = out;
  t0 = now();
  for (int i = 0; i < 64; ++i) { double x = gp[(idx + i * 7) & 255]; idx = ((int)x + idx) & 255; }
  t1 = now(); if (tid == 0) t[9] = (t1 - t0) / 64; out[2048 + tid] = idx;
}
int main() {
  double* out; unsigned long long* t; hipMalloc(&out, 4096 * 8); hipMalloc(&t, 16 * 8); hipMemset(out, 0, 4096 * 8);
  probe<<<1, 256>>>(out, t, 1.0); hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[] = {"dependent ds_read", "dependent v_fma_f64", "__syncthreads (4 waves)", "rsq + 2 Newton + corrections", "dependent mfma_f64_4x4x4",
                      "LDS write -> barrier -> read -> barrier hop", "in-wave LDS write -> fence -> read hop", "f64 division", "DPP quad bcast (f64) + add", "dependent global load (L2)"};
  for (int i = 0; i < 10; ++i) printf("%-48s %llu cycles\n", nm[i], h[i]);
  return 0;
}
