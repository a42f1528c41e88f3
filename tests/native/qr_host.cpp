// CPU build of the per-lane spectral-radius routine of tzddpc_amd/csrc/tz_gain.hip.h (the device qualifiers compiled away): the
// algorithm (Hessenberg reduction + Francis QR with deflation and exceptional shifts) is checked against LAPACK without a GPU, and
// can be run under -fsanitize=address,undefined (tests/test_host_qr.py).  Test infrastructure only.
#include <algorithm>
#include <cmath>
#include <cstdio>
#define __device__
#define __global__
#define __shared__ static
#define __launch_bounds__(x)
using std::max;
using std::min;
struct dim3x { int x; };
static dim3x threadIdx, blockIdx, blockDim;
static void __syncthreads() {}
static bool __any(bool b) { return b; }
static int __syncthreads_or(int v) { return v; }
enum { RED_SUM = 0 };
template <int OP> static double tz_wave_reduce(double v) { return v; }      // (the adversary kernel is compiled, never run, on the host)
#include "../../tzddpc_amd/csrc/tz_gain.hip.h"

extern "C" int specrad_host(int S, int n, const double* Ms, double* rho) {
  static double slab[64 * 64];
  int bad = 0;
  for (int s = 0; s < S; ++s) {
    for (int e = 0; e < n * n; ++e) slab[e << 6] = Ms[(size_t)s * n * n + e];
    if (!tz_spectral_radius(slab, n, rho[s])) ++bad;
  }
  return bad;
}
