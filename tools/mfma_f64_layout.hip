// Empirical lane-layout probe for the f64 MFMA shapes on gfx950 (guide: "check the map with exact
// integer data before relying on it").  One-hot A lane x uniquely-valued B lanes -> D.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe4(double* out) {   // out[64 runs][64 lanes]
  int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    double a = (lane == la) ? 1.0 : 0.0;
    double b = 1.0 + lane;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[la * 64 + lane] = d;
  }
}
__global__ void probe16(double* out) {  // out[64 runs][64 lanes][4 regs]
  int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    double a = (lane == la) ? 1.0 : 0.0;
    double b = 1.0 + lane;
    d4 c = {0, 0, 0, 0};
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(la * 64 + lane) * 4 + r] = d[r];
  }
}
int main() {
  double *d4o, *d16o; hipMalloc(&d4o, 64 * 64 * 8); hipMalloc(&d16o, 64 * 64 * 4 * 8);
  probe4<<<1, 64>>>(d4o); probe16<<<1, 64>>>(d16o); hipDeviceSynchronize();
  static double h4[64 * 64], h16[64 * 64 * 4];
  hipMemcpy(h4, d4o, sizeof(h4), hipMemcpyDeviceToHost); hipMemcpy(h16, d16o, sizeof(h16), hipMemcpyDeviceToHost);
  printf("== 4x4x4_4b: for A one-hot at lane la: list of (out_lane <- b_lane)\n");
  for (int la = 0; la < 64; ++la) {
    printf("la %2d:", la);
    for (int l = 0; l < 64; ++l) if (h4[la * 64 + l] != 0) printf(" %d<-%d", l, (int)h4[la * 64 + l] - 1);
    printf("\n");
  }
  printf("== 16x16x4: la: (out_lane,reg <- b_lane)\n");
  for (int la = 0; la < 64; ++la) {
    printf("la %2d:", la);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (h16[(la * 64 + l) * 4 + r] != 0) printf(" %d.%d<-%d", l, r, (int)h16[(la * 64 + l) * 4 + r] - 1);
    printf("\n");
  }
  return 0;
}
