// Device kernels of the TZDDPC hot path for gfx950 (MI355X).  Included by tzddpc_hip.hip only.
//
//   tz_tube_kernel     numeric tube propagation from e0   (reference tzddpc/tzddpc.py:172-181,191-192)
//   tz_affine_kernel   q(theta), h(theta), parameter rows  (cvxpy's parameter application at :364-365)
//   tz_ipm_kernel      batched Mehrotra interior point     (problem_full.solve at :367)
//   tz_finish_kernel   v, xbar, objective, active set      (return tuple at :377)
//   tz_plant_kernel    closed-loop plant / error update    (examples/1.double_integrator_sim.py:83-87)
//
// f64 everywhere.  The only matrix-core instruction used is v_mfma_f64_4x4x4_4b_f64, measured on
// MI355X at 16 cycles/instruction/SIMD (~73 TFLOP/s chip-wide, tools/mfma_f64_rate.hip) while
// v_mfma_f64_16x16x4 tops out at 35-46 TFLOP/s; lane layout (probed, tools/mfma_f64_layout.hip):
//     A_blk[i][k] : lane 16k + 4blk + i      B_blk[k][j] : lane 16k + 4blk + j
//     D_blk[i][j] : lane 16i + 4blk + j      (4 independent 4x4x4 products, blk = 0..3)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TZ_THREADS 256
#define TZ_NWAVES 4
#define TZ_NMAX 8          // max dim_x supported by the tube kernel's register arrays
#define TZ_MMAX 4

struct TzCsr {
  int rows;
  const int* ptr;
  const int* col;
  const double* val;
  const double* c0;
};

// ------------------------------------------------------------------------------------------------
// Tube propagation: one thread per trajectory.
//   Z^(0) = <e0>;  Z^(p+1) = M_K Z^(p):  c^(p+1) = C_K c^(p),  beta_p = D_K (|c^(p)| + rad^(p)),
//   rad^(p+1) = sum_{l<=p} |C_K^(p-l)| beta_l,   radU^(p+1) = sum_{l<=p} |K C_K^(p-l)| beta_l
// theta = [xbar0 | |xbar0| | (c_k, rho^x_k, rho^u_k) for k < N] with step k taking power[k].
// ------------------------------------------------------------------------------------------------
struct TubeParams {
  int B, n, m, N, pmax, ntheta;
  const double* CK; const double* DK;
  const double* absCK;   // pmax x n x n
  const double* absKCK;  // pmax x m x n
  const int* power;      // N
  const double* xbar0; const double* e0;   // B x n
  double* theta;         // B x ntheta
  int* prestatus;        // B: cleared here, set by tz_affine_kernel when a parameter-only row is violated
  double* ws;            // B x (pmax+1) x (3n + m): [c | beta | radx | radu] per power (only when use_lds == 0)
  int use_lds;
};

// One wave per trajectory.  Lane l keeps beta_l (and beta_{l+64}, ...) in registers; the radius sums over the history
// are wave reductions, so the recursion costs O(pmax) reduction steps instead of an O(pmax^2) dependent chain per thread.
#define TZ_TUBE_SLOTS 2      // history entries per lane: pmax <= 64 * TZ_TUBE_SLOTS
__device__ inline double tz_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__global__ __launch_bounds__(64) void tz_tube_kernel(TubeParams p) {
  __shared__ double hist[(64 * TZ_TUBE_SLOTS + 1) * (2 * TZ_NMAX + TZ_MMAX)];     // per power: c | radx | radu
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = p.n, m = p.m;
  const int hs = 2 * n + m;
  double c[TZ_NMAX], rx[TZ_NMAX], ru[TZ_MMAX], beta[TZ_TUBE_SLOTS][TZ_NMAX];
#pragma unroll
  for (int i = 0; i < TZ_NMAX; ++i) { c[i] = (i < n) ? p.e0[(size_t)b * n + i] : 0.0; rx[i] = 0.0; }
#pragma unroll
  for (int j = 0; j < TZ_MMAX; ++j) ru[j] = 0.0;
#pragma unroll
  for (int sl = 0; sl < TZ_TUBE_SLOTS; ++sl)
#pragma unroll
    for (int i = 0; i < TZ_NMAX; ++i) beta[sl][i] = 0.0;
  if (lane == 0) { for (int i = 0; i < n; ++i) { hist[i] = c[i]; hist[n + i] = 0.0; } for (int j = 0; j < m; ++j) hist[2 * n + j] = 0.0; }
  for (int pw = 0; pw < p.pmax; ++pw) {
    // beta_pw = DK (|c| + radx)  (uniform), kept by lane pw % 64 in slot pw / 64
    double bnew[TZ_NMAX], cn[TZ_NMAX];
#pragma unroll
    for (int i = 0; i < TZ_NMAX; ++i) {
      double acc = 0.0, acc2 = 0.0;
      if (i < n) for (int j = 0; j < n; ++j) { acc += p.DK[i * n + j] * (fabs(c[j]) + rx[j]); acc2 += p.CK[i * n + j] * c[j]; }
      bnew[i] = acc; cn[i] = acc2;
    }
#pragma unroll
    for (int sl = 0; sl < TZ_TUBE_SLOTS; ++sl)
      if (lane + 64 * sl == pw) {
#pragma unroll
        for (int i = 0; i < TZ_NMAX; ++i) beta[sl][i] = bnew[i];
      }
    // rad^(pw+1) = sum_{l <= pw} |C^(pw-l)| beta_l ,  radU^(pw+1) = sum_l |K C^(pw-l)| beta_l
    double px[TZ_NMAX], pu[TZ_MMAX];
#pragma unroll
    for (int i = 0; i < TZ_NMAX; ++i) px[i] = 0.0;
#pragma unroll
    for (int j = 0; j < TZ_MMAX; ++j) pu[j] = 0.0;
#pragma unroll
    for (int sl = 0; sl < TZ_TUBE_SLOTS; ++sl) {
      const int l = lane + 64 * sl;
      if (l <= pw) {
        const double* Mx = p.absCK + (size_t)(pw - l) * n * n;
        const double* Mu = p.absKCK + (size_t)(pw - l) * m * n;
#pragma unroll
        for (int i = 0; i < TZ_NMAX; ++i) if (i < n) for (int j = 0; j < n; ++j) px[i] += Mx[i * n + j] * beta[sl][j];
#pragma unroll
        for (int i = 0; i < TZ_MMAX; ++i) if (i < m) for (int j = 0; j < n; ++j) pu[i] += Mu[i * n + j] * beta[sl][j];
      }
    }
#pragma unroll
    for (int i = 0; i < TZ_NMAX; ++i) { if (i < n) rx[i] = tz_wave_sum(px[i]); c[i] = cn[i]; }
#pragma unroll
    for (int j = 0; j < TZ_MMAX; ++j) if (j < m) ru[j] = tz_wave_sum(pu[j]);
    if (lane == 0) {
      double* h = hist + (size_t)(pw + 1) * hs;
      for (int i = 0; i < n; ++i) { h[i] = c[i]; h[n + i] = rx[i]; }
      for (int j = 0; j < m; ++j) h[2 * n + j] = ru[j];
    }
  }
  __syncthreads();
  double* th = p.theta + (size_t)b * p.ntheta;
  if (lane < n) { const double x0 = p.xbar0[(size_t)b * n + lane]; th[lane] = x0; th[n + lane] = fabs(x0); }
  if (lane == 0) p.prestatus[b] = 0;
  for (int e = lane; e < p.N * hs; e += 64) {
    const int k = e / hs, idx = e % hs;
    th[2 * n + e] = hist[(size_t)p.power[k] * hs + idx];
  }
}

// ------------------------------------------------------------------------------------------------
// Affine maps over theta: one thread per (trajectory, row); rows = [q (nz) | h (mi) | par (npar)].
// ------------------------------------------------------------------------------------------------
struct AffineParams {
  int B, ntheta, nz, mi, npar;
  TzCsr q, h, par;
  const double* par_lo; const double* par_hi;
  const double* theta;
  double* qv; double* hv;   // B x nz, B x mi
  int* prestatus;           // B (zeroed before launch); set to 1 when a parameter row is violated
};

__device__ inline double csr_row(const TzCsr& M, int r, const double* th) {
  double acc = M.c0[r];
  for (int e = M.ptr[r]; e < M.ptr[r + 1]; ++e) acc += M.val[e] * th[M.col[e]];
  return acc;
}

__global__ void tz_affine_kernel(AffineParams p) {
  const int rows = p.nz + p.mi + p.npar;
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)p.B * rows) return;
  int b = (int)(gid / rows), r = (int)(gid % rows);
  const double* th = p.theta + (size_t)b * p.ntheta;
  if (r < p.nz) {
    p.qv[(size_t)b * p.nz + r] = csr_row(p.q, r, th);
  } else if (r < p.nz + p.mi) {
    int rr = r - p.nz;
    p.hv[(size_t)b * p.mi + rr] = csr_row(p.h, rr, th);
  } else {
    int rr = r - p.nz - p.mi;
    double v = csr_row(p.par, rr, th);
    const double tolp = 1e-9;
    if (!(v >= p.par_lo[rr] - tolp) || !(v <= p.par_hi[rr] + tolp)) p.prestatus[b] = 1;
  }
}

#include "tz_ipm.hip.h"

// ------------------------------------------------------------------------------------------------
// Finish: one wave per trajectory.
// ------------------------------------------------------------------------------------------------
struct FinishParams {
  int B, n, m, N, nz, mi, nzp, nc_rows;
  const double* P;  const double* Dz; const double* Phi; const double* Gam;
  const double* r1; const double* R2; double r0; double cost_scale;
  const int* row_of; const double* act_scale;
  const double* xbar0; const double* q; const double* x; const double* s; const double* lam;
  const int* status;
  double* v; double* xbar; double* cost; uint8_t* active;   // active may be null
  size_t cost_stride;
};

__global__ __launch_bounds__(64) void tz_finish_kernel(FinishParams p) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nz = p.nz, n = p.n, m = p.m, N = p.N;
  const double* x = p.x + (size_t)b * nz;
  const double* q = p.q + (size_t)b * nz;
  const double* x0 = p.xbar0 + (size_t)b * n;
  const int nv = N * m;
  // objective (scaled space) = sum_c x_c (0.5 (P x)_c + q_c)
  double acc = 0.0;
  for (int c = lane; c < nz; c += 64) {
    double px = 0.0;
    const double* row = p.P + (size_t)c * p.nzp;
    for (int k = 0; k < nz; ++k) px += row[k] * x[k];
    acc += x[c] * (0.5 * px + q[c]);
  }
  acc = tz_wave_reduce<RED_SUM>(acc);
  if (lane == 0) {
    double r = p.r0;
    for (int i = 0; i < n; ++i) {
      r += p.r1[i] * x0[i];
      for (int j = 0; j < n; ++j) r += x0[i] * p.R2[i * n + j] * x0[j];
    }
    const int st = p.status[b];
    p.cost[(size_t)b * p.cost_stride] = (st == 0 || st == 1) ? acc / p.cost_scale + r : INFINITY;
  }
  double* v = p.v + (size_t)b * nv;
  for (int c = lane; c < nv; c += 64) v[c] = p.Dz[c] * x[c];
  double* xb = p.xbar + (size_t)b * (N + 1) * n;
  for (int r = lane; r < (N + 1) * n; r += 64) {
    double a = 0.0;
    for (int j = 0; j < n; ++j) a += p.Phi[(size_t)r * n + j] * x0[j];
    const double* g = p.Gam + (size_t)r * nv;
    for (int c = 0; c < nv; ++c) a += g[c] * (p.Dz[c] * x[c]);
    xb[r] = a;
  }
  if (p.active) {
    uint8_t* act = p.active + (size_t)b * p.nc_rows;
    for (int r = lane; r < p.nc_rows; r += 64) act[r] = 0;
    __syncthreads();
    const double* s = p.s + (size_t)b * p.mi;
    const double* lam = p.lam + (size_t)b * p.mi;
    for (int r = lane; r < p.mi; r += 64)
      if (s[r] * p.act_scale[r] < lam[r]) act[p.row_of[r]] = 1;
  }
}

// ------------------------------------------------------------------------------------------------
// Plant / error update: one thread per trajectory.
//   u = K e + v0 ; x+ = A x + B u + w ; xbar+ = xbar[1] ; e+ = x+ - xbar+
// ------------------------------------------------------------------------------------------------
struct PlantParams {
  int B, n, m, N;
  const double* K; const double* A; const double* Bm;
  const double* v; const double* xbar_pred;   // B x N x m, B x (N+1) x n
  const double* w;                            // B x n  (stride w_stride between trajectories)
  size_t w_stride;
  const int* status;
  double* x; double* xbar; double* e;         // B x n in/out
  double* u_out; size_t u_stride;             // may be null
  double* x_out; size_t x_stride;             // may be null: copy of x+
  int* sticky;                                // may be null: first non-zero status kept
};

__global__ void tz_plant_kernel(PlantParams p) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  const int n = p.n, m = p.m;
  double x[TZ_NMAX], e[TZ_NMAX], u[TZ_MMAX], xn[TZ_NMAX];
  const int st = p.status[b];
  if (p.sticky && p.sticky[b] == 0 && st != 0) p.sticky[b] = st;
  for (int i = 0; i < n; ++i) { x[i] = p.x[(size_t)b * n + i]; e[i] = p.e[(size_t)b * n + i]; }
  for (int j = 0; j < m; ++j) {
    double a = p.v[(size_t)b * p.N * m + j];
    for (int i = 0; i < n; ++i) a += p.K[j * n + i] * e[i];
    u[j] = a;
    if (p.u_out) p.u_out[(size_t)b * p.u_stride + j] = a;
  }
  for (int i = 0; i < n; ++i) {
    double a = p.w[(size_t)b * p.w_stride + i];
    for (int j = 0; j < n; ++j) a += p.A[i * n + j] * x[j];
    for (int j = 0; j < m; ++j) a += p.Bm[i * m + j] * u[j];
    xn[i] = a;
  }
  for (int i = 0; i < n; ++i) {
    const double xb = p.xbar_pred[(size_t)b * (p.N + 1) * n + n + i];
    p.x[(size_t)b * n + i] = xn[i];
    p.xbar[(size_t)b * n + i] = xb;
    p.e[(size_t)b * n + i] = xn[i] - xb;
    if (p.x_out) p.x_out[(size_t)b * p.x_stride + i] = xn[i];
  }
}
