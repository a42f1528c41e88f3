// Micro-benchmark of the k-split Gram loop (tz_form_H_ksplit): where do the cycles go?  One or more workgroups of 256 threads.
//   hipcc --offload-arch=gfx950 -O3 -o gram_probe tools/gram_probe.hip && ./gram_probe [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define TZ 10
#define KC 64
__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
__device__ inline double ld_pinned(const double* p) {
  unsigned long long u = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  return __builtin_bit_cast(double, u);
}
__device__ inline double sel4(int blk, double v0, double v1, double v2, double v3) {
  const double a = (blk & 1) ? v1 : v0, b = (blk & 1) ? v3 : v2;
  return (blk & 2) ? b : a;
}
struct Stage { double v[TZ]; double w; };

// MODE 0: loads + mfma (2 stages), 1: loads only, 2: mfma only, 3: scalar (readfirstlane) wave index, 3 stages
template <int MODE>
__global__ __launch_bounds__(256) void gram(const double* Gp, double* out, unsigned long long* t, int reps) {
  __shared__ double wv[4 * KC + 8];
  const int lane = threadIdx.x & 63;
  const int wave = (MODE == 3) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (threadIdx.x >> 6);
  const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
  for (int i = threadIdx.x; i < 4 * KC + 8; i += 256) wv[i] = (i < 4 * KC) ? 1.0 + 1e-3 * i : 0.0;
  __syncthreads();
  const unsigned rowbytes = (TZ + 1) * 128u, laneoff = (4 * k + ij) * 8u;
  const char* gp = (const char*)Gp;
  double acc[TZ][3];
#pragma unroll
  for (int I = 0; I < TZ; ++I) for (int q = 0; q < 3; ++q) acc[I][q] = 0.0;
  auto load = [&](int kc, Stage& st) {
    const int kcc = kc < KC ? kc : KC;
    st.w = wv[4 * kcc + k];
    const char* prow = gp + (size_t)kcc * rowbytes + laneoff;
#pragma unroll
    for (int J = 0; J < TZ; ++J) st.v[J] = ld_pinned((const double*)(prow + J * 128u));
  };
  auto mma = [&](const Stage& st) {
    if (MODE == 1) { for (int J = 0; J < TZ; ++J) acc[J][0] += st.v[J] * st.w; return; }
    const double b0 = sel4(blk, st.v[0], st.v[1], st.v[2], st.v[3]);
    const double b1 = sel4(blk, st.v[4], st.v[5], st.v[6], st.v[7]);
    const double b2 = sel4(blk, st.v[8], st.v[9], 0.0, 0.0);
#pragma unroll
    for (int I = 0; I < TZ; ++I) {
      const double a = st.v[I] * st.w;
      acc[I][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b0, acc[I][0], 0, 0, 0);
      if (I >= 4) acc[I][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b1, acc[I][1], 0, 0, 0);
      if (I >= 8) acc[I][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b2, acc[I][2], 0, 0, 0);
    }
  };
  unsigned long long t0 = now();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 2) {
      Stage s0; for (int J = 0; J < TZ; ++J) s0.v[J] = 1.0 + lane * 1e-3 + J; s0.w = 0.5;
      for (int kc = wave; kc < KC; kc += 4) { mma(s0); s0.w += 1e-9; }
    } else if (MODE == 3) {
      Stage s0, s1, s2;
      load(wave, s0); load(wave + 4, s1);
      for (int kc = wave; kc < KC; kc += 12) {
        load(kc + 8, s2); mma(s0);
        load(kc + 12, s0); mma(s1);
        load(kc + 16, s1); mma(s2);
      }
    } else {
      Stage s0, s1;
      load(wave, s0);
      for (int kc = wave; kc < KC; kc += 8) {
        load(kc + 4, s1); mma(s0);
        load(kc + 8, s0); mma(s1);
      }
    }
  }
  unsigned long long t1 = now();
  if (lane == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = (t1 - t0) / reps;
  double sacc = 0.0;
#pragma unroll
  for (int I = 0; I < TZ; ++I) for (int q = 0; q < 3; ++q) sacc += acc[I][q];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
}

// dependent load chain over a footprint of `bytes` (stride one cache line), one lane
__global__ void chase(const int* next, int* out, unsigned long long* t, int n) {
  int idx = threadIdx.x;
  unsigned long long t0 = now();
  for (int i = 0; i < n; ++i) idx = next[idx * 32];
  unsigned long long t1 = now();
  if (threadIdx.x == 0) { t[0] = (t1 - t0) / n; out[0] = idx; }
}

int main(int argc, char** argv) {
  int blocks = argc > 1 ? atoi(argv[1]) : 1;
  size_t n = (size_t)(KC + 1) * (TZ + 1) * 16;
  std::vector<double> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = 1e-3 * (double)(i % 97);
  double *Gp, *out; unsigned long long* t;
  hipMalloc(&Gp, n * 8); hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&t, (size_t)blocks * 4 * 8 + 64);
  hipMemcpy(Gp, h.data(), n * 8, hipMemcpyHostToDevice);
  const char* nm[] = {"loads + mfma, 2 stages", "loads only", "mfma only", "scalar wave id, 3 stages"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int pass = 0; pass < 2; ++pass) {
      if (mode == 0) gram<0><<<blocks, 256>>>(Gp, out, t, 8);
      if (mode == 1) gram<1><<<blocks, 256>>>(Gp, out, t, 8);
      if (mode == 2) gram<2><<<blocks, 256>>>(Gp, out, t, 8);
      if (mode == 3) gram<3><<<blocks, 256>>>(Gp, out, t, 8);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> ht((size_t)blocks * 4);
    hipMemcpy(ht.data(), t, ht.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0, sum = 0; for (auto v : ht) { mx = v > mx ? v : mx; sum += v; }
    printf("%-28s blocks %d: wave0 %llu  mean %llu  max %llu cycles per Gram (16 k-steps per wave)\n", nm[mode], blocks, ht[0], sum / ht.size(), mx);
  }
  // pointer chase: footprints 16 KB (L1), 256 KB (L2), 64 MB (beyond L2)
  for (size_t bytes : {16u << 10, 256u << 10, 64u << 20}) {
    size_t lines = bytes / 128; std::vector<int> nx(lines * 32, 0);
    for (size_t i = 0; i < lines; ++i) nx[i * 32] = (int)((i * 769 + 13) % lines);
    int* dn; int* dout; hipMalloc(&dn, nx.size() * 4); hipMalloc(&dout, 64);
    hipMemcpy(dn, nx.data(), nx.size() * 4, hipMemcpyHostToDevice);
    chase<<<1, 1>>>(dn, dout, t, 2000); hipDeviceSynchronize();
    chase<<<1, 1>>>(dn, dout, t, 2000); hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, t, 8, hipMemcpyDeviceToHost);
    printf("dependent global load, footprint %zu KB: %llu cycles\n", bytes >> 10, c);
    hipFree(dn); hipFree(dout);
  }
  return 0;
}
