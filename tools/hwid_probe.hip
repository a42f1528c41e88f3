// tools/hwid_probe.hip: where the hardware puts the waves of a 1024 x 256-thread launch shaped like tz_ipm_kernel (39 KB LDS, 4 workgroups
// per CU): per workgroup the XCC, SE, CU and the SIMD of each of its four waves.  hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
__global__ __launch_bounds__(256, 4) void probe(unsigned* out, int spin) {
  __shared__ double pad[4900];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  pad[threadIdx.x] = hw;
  __syncthreads();
  double acc = pad[(threadIdx.x * 7) & 255];
  for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 1e-9;   // keep the workgroups resident together
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 0] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (acc == 12345.0);
  }
}
int main() {
  const int B = 1024;
  unsigned* d; hipMalloc(&d, B * 4 * 2 * 4);
  std::vector<unsigned> h(B * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(B), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d, B * 32, hipMemcpyDeviceToHost);
  std::map<std::tuple<int,int,int,int>, std::vector<int>> cu;   // (xcc, se, sh, cu) -> workgroups
  int same = 0;
  for (int b = 0; b < B; ++b) {
    unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
    int cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cu[{(int)xcc, se, sh, cuid}].push_back(b);
  }
  printf("distinct CUs %zu\n", cu.size());
  int shown = 0; std::map<int,int> hist;            // how many of a CU's workgroups have wave 0 on the same SIMD
  std::map<int,int> perwg;                          // SIMD pattern of the four waves of a workgroup
  for (auto& kv : cu) {
    int cnt[4] = {0, 0, 0, 0};
    for (int b : kv.second) cnt[(h[b * 8] >> 4) & 3]++;
    int mx = 0; for (int s = 0; s < 4; ++s) mx = cnt[s] > mx ? cnt[s] : mx;
    hist[mx * 10 + (int)kv.second.size()]++;
    if (shown++ < 6) {
      printf("xcc %d se %d sh %d cu %2d:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
      for (int b : kv.second) { printf("  wg %4d simd", b); for (int w = 0; w < 4; ++w) printf(" %u", (h[(b * 4 + w) * 2] >> 4) & 3); }
      printf("\n");
    }
  }
  for (int b = 0; b < B; ++b) { int pat = 0; for (int w = 0; w < 4; ++w) pat = pat * 10 + ((h[(b * 4 + w) * 2] >> 4) & 3); perwg[pat]++; }
  for (auto& kv : hist) printf("CUs with %d workgroups of which %d share wave 0's SIMD: %d\n", kv.first % 10, kv.first / 10, kv.second);
  for (auto& kv : perwg) printf("wave->SIMD pattern %04d: %d workgroups\n", kv.first, kv.second);
  return 0;
}
