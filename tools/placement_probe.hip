// Where do the four waves of a 256-thread workgroup land?  Same launch shape as tz_ipm_kernel on the bench workload (1024 workgroups,
// ~36 KB of LDS each, all resident at once): every wave records HW_ID (SIMD, CU, SE, wave slot) and XCC_ID.
// hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o tools/bin/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 4) void probe(unsigned* out, int spin) {
  extern __shared__ double lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = hw;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);      // bounded: keeps the grid resident while the rest is dispatched
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = hw; out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = xcc; }
}
int main() {
  const int B = 1024;
  unsigned* d; hipMalloc(&d, B * 8 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024);
  hipLaunchKernelGGL(probe, dim3(B), dim3(256), 36 * 1024, 0, d, 2000);
  std::vector<unsigned> h(B * 8);
  if (hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] tg_id[19:16] vm_id[23:20] queue_id[26:24] state_id[29:27] me_id[31:30]
  int simd_of_wave[4][4] = {};
  std::map<unsigned, std::vector<int>> cu_blocks;                  // (xcc, se, sh, cu) -> blocks
  for (int b = 0; b < B; ++b) {
    for (int w = 0; w < 4; ++w) { unsigned hw = h[2 * (b * 4 + w)]; simd_of_wave[w][(hw >> 4) & 3]++; }
    unsigned hw = h[2 * (b * 4)], xcc = h[2 * (b * 4) + 1] & 0xf;
    cu_blocks[(xcc << 16) | (hw & 0xff00)].push_back(b);
  }
  printf("SIMD of wave w (rows: wave 0..3, columns: SIMD 0..3), over %d workgroups:\n", B);
  for (int w = 0; w < 4; ++w) printf("  wave %d: %5d %5d %5d %5d\n", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
  printf("distinct CUs used: %zu\n", cu_blocks.size());
  int shown = 0;
  for (auto& kv : cu_blocks) {
    if (shown++ >= 12) break;
    printf("  xcc %u se %u sh %u cu %2u: blocks", kv.first >> 16, (kv.first >> 13) & 7, (kv.first >> 12) & 1, (kv.first >> 8) & 15);
    for (int b : kv.second) { printf(" %4d[simd", b); for (int w = 0; w < 4; ++w) printf("%u", (h[2 * (b * 4 + w)] >> 4) & 3); printf(" slot"); for (int w = 0; w < 4; ++w) printf("%u", h[2 * (b * 4 + w)] & 15); printf("]"); }
    printf("\n");
  }
  std::map<size_t, int> hist; for (auto& kv : cu_blocks) hist[kv.second.size()]++;
  for (auto& kv : hist) printf("CUs holding %zu workgroups: %d\n", kv.first, kv.second);
  return 0;
}
