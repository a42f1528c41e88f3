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

#ifndef TZ_PROFILE
#define TZ_PROFILE 0
#endif
#define TZ_THREADS 256
#define TZ_NWAVES 4
#define TZ_NMAX 16         // max dim_x / dim_u of the MPC path (register arrays of the plant update, LDS slots of the closed-loop state);
#define TZ_MMAX 8          // K0 (tz_identify.hip.h) and the gain kernels (tz_gain.hip.h) keep their own limits of 8 / 4

// Affine map rows over theta in ELL form: entry e of row r at [e * rows + r] (coalesced over rows), W entries per row, rows
// with fewer non-zeros padded with (0.0, column 0).  No row pointers: every load of a row is independent of the others.  An entry
// is one 16-byte record (value, byte offset of the theta entry, zero): one vector-memory instruction per non-zero.
struct __attribute__((aligned(16))) TzEllEnt { double val; unsigned off; unsigned pad; };
typedef double tz_d2 __attribute__((ext_vector_type(2)));
// byte offset of the input entry: both halves of the record's second double are used (the second is zero), so that the record stays
// ONE 16-byte load (the compiler splits a load of which only 12 bytes are consumed into two); base + off + pad is one v_add3_u32
__device__ inline unsigned tz_ell_off(double y) {
  const unsigned long long w = __builtin_bit_cast(unsigned long long, y);
  return (unsigned)w + (unsigned)(w >> 32);
}
struct TzCsr {
  int rows, W;
  const TzEllEnt* ent;
  const double* c0;
};

// ------------------------------------------------------------------------------------------------
// Tube propagation from e0 (reference tzddpc/tzddpc.py:172-181,191-192).
//   Z^(0) = <e0>;  Z^(p+1) = M_K Z^(p):  c^(p+1) = C_K c^(p),  beta_p = D_K (|c^(p)| + rad^(p)),
//   rad^(p+1) = sum_{l<=p} |C_K^(p-l)| beta_l,   radU^(p+1) = sum_{l<=p} |K C_K^(p-l)| beta_l
// Given a_l = |c^(l)| the recursion is linear with nonnegative coefficients, so its resolvent is a constant of the
// problem (block lower-triangular Toeplitz, built once on the host in tz_problem_create):
//   rad^(p) = sum_{l<p} Tx_{p-1-l} a_l ,  radU^(p) = sum_{l<p} Tu_{p-1-l} a_l ,  c^(l) = C_K^l e0
// which turns an O(pmax) dependent chain into two short parallel passes.
// theta = [xbar0 | |xbar0| | (c_k, rho^x_k, rho^u_k) for k < N] with step k taking power[k].
// ------------------------------------------------------------------------------------------------
struct TubeParams {
  int B, n, m, N, pmax, ntheta;
  const double* CKpow;   // (pmax+1) x n x n      C_K^l
  const double* T;       // pmax x (n+m) x n      [Tx_d ; Tu_d]
  const int* power;      // N
  const double* xbar0; const double* e0;   // B x n
  double* theta;         // B x ntheta
  int* prestatus;        // B: cleared here, set by tz_affine_kernel when a parameter-only row is violated
};

#define TZ_PMAX 128        // highest supported power of M_K
template <int CTRL>
__device__ inline double tz_quad_xor(double v) {                       // DPP quad_perm of a double (0xB1: lane^1, 0x4E: lane^2)
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xf, 0xf, true);        // (no `old` operand to initialise)
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// All nt threads of the workgroup (tid) work on trajectory b.  aL: pmax * n doubles of LDS scratch; th: ntheta doubles (LDS or
// global).  Two stages with a workgroup barrier between them (tz_tube_block); the caller adds another before th is read by other
// threads.  NC, MC: dim_x, dim_u known at compile time (0: taken from p) -- the index divisions become shifts / multiplies and the dot
// products unroll; power: p.power or a copy of it in LDS.
template <int NC = 0, int MC = 0>
__device__ inline void tz_tube_stage1(const TubeParams& p, const double* CKpow, const double* xbar0, const double* e0, double* aL, double* th, int tid, int nt) {
  const int n = NC ? NC : p.n;
  for (int e = tid; e < p.pmax * n; e += nt) {
    const double* M = CKpow + (size_t)e * n;              // row i of C_K^l with e = l n + i
    double a = 0.0;
    for (int j = 0; j < n; ++j) a += M[j] * e0[j];
    aL[e] = fabs(a);
  }
  if (tid < n) { const double x0 = xbar0[tid]; th[tid] = x0; th[n + tid] = fabs(x0); }
}
template <int NC = 0, int MC = 0>
__device__ inline void tz_tube_stage2(const TubeParams& p, const double* CKpow, const double* Ttab, const int* power, const double* e0, const double* aL, double* th, int tid, int nt) {
  const int n = NC ? NC : p.n, m = MC ? MC : p.m, hs = 2 * n + m;
  for (int e = tid; e < p.N * n; e += nt) {                      // centres c_k = C_K^power[k] e0
    const int k = e / n, i = e - k * n;
    const double* M = CKpow + ((size_t)power[k] * n + i) * n;
    double a = 0.0;
    for (int j = 0; j < n; ++j) a += M[j] * e0[j];
    th[2 * n + k * hs + i] = a;
  }
  // radii: four lanes per entry (history index l = sub, sub + 4, ...), folded with two quad permutes
  const int sub = tid & 3;
  for (int q = tid >> 2; q < p.N * (n + m); q += nt >> 2) {
    const int k = q / (n + m), comp = q - k * (n + m), pw = power[k];
    double a = 0.0;
    for (int l = sub; l < pw; l += 4) {
      const double* T = Ttab + ((size_t)(pw - 1 - l) * (n + m) + comp) * n;
      const double* al = aL + l * n;
      for (int j = 0; j < n; ++j) a += T[j] * al[j];
    }
    a += tz_quad_xor<0xB1>(a);
    a += tz_quad_xor<0x4E>(a);
    if (sub == 0) th[2 * n + k * hs + n + comp] = a;
  }
}
template <int NC = 0, int MC = 0>
__device__ inline void tz_tube_block(const TubeParams& p, const double* CKpow, const double* Ttab, const int* power, const double* xbar0, const double* e0, double* aL, double* th, int tid, int nt) {
  tz_tube_stage1<NC, MC>(p, CKpow, xbar0, e0, aL, th, tid, nt);
  __syncthreads();
  tz_tube_stage2<NC, MC>(p, CKpow, Ttab, power, e0, aL, th, tid, nt);
}

__global__ __launch_bounds__(64) void tz_tube_kernel(TubeParams p) {
  __shared__ double aL[TZ_PMAX * TZ_NMAX];
  const int b = blockIdx.x;
  tz_tube_block(p, p.CKpow, p.T, p.power, p.xbar0 + (size_t)b * p.n, p.e0 + (size_t)b * p.n, aL, p.theta + (size_t)b * p.ntheta, threadIdx.x, 64);
  if (threadIdx.x == 0) p.prestatus[b] = 0;
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

#define TZ_ELL_REG 8
__device__ inline double csr_row(const TzCsr& M, int r, const double* th) {
  double acc = M.c0[r];
  const char* tb = reinterpret_cast<const char*>(th);
  tz_d2 v[TZ_ELL_REG];
#pragma unroll
  for (int e = 0; e < TZ_ELL_REG; ++e) { v[e] = tz_d2{0.0, 0.0}; if (e < M.W) v[e] = *reinterpret_cast<const tz_d2*>(M.ent + (size_t)e * M.rows + r); }
#pragma unroll
  for (int e = 0; e < TZ_ELL_REG; ++e) if (e < M.W) acc += v[e].x * *reinterpret_cast<const double*>(tb + tz_ell_off(v[e].y));
  for (int e = TZ_ELL_REG; e < M.W; ++e) {
    const tz_d2 q = *reinterpret_cast<const tz_d2*>(M.ent + (size_t)e * M.rows + r);
    acc += q.x * *reinterpret_cast<const double*>(tb + tz_ell_off(q.y));
  }
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

struct FinishParams {
  int B, n, m, N, nz, mi, nzp, nc_rows;
  const double* P;  const double* Dz; const double* Phi; const double* Gam;
  const double* r1; const double* R2; double r0; double cost_scale;
  const int* row_of; const double* act_scale;
  const double* xbar0; const double* q; const double* x; const double* s; const double* lam;
  const int* status;
  double* v; double* xbar; double* cost; uint8_t* active;   // active may be null
  size_t cost_stride;
  const int* vpos;       // device position of v[k, j] in x (staircase ordering of the library); null = identity
  const double* rec0; const double* recx; const double* recy;   // equality-eliminated problems: v = rec0 + recx xbar0 + recy x (recy in device column order), else null
};

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

// Everything the fused closed-loop step needs besides the interior point itself (tz_ipm_kernel with F.on != 0 does
// tube -> parameter maps -> solve -> recovery / objective -> plant update in one launch; theta, q, h stay in LDS / registers).
struct FuseParams {
  int on;
  int lean_epilogue;                // xbar[1] depends on v[0] only and v is a scaled copy of x (no equality elimination): the recovery and the plant
                                    // update of a step that reports no cost / v / xbar are done by one wave without workgroup barriers
  int sticky_fresh;                 // the launch starts a run: plant.sticky is cleared by the kernel (no memset dispatch in front of it)
  int npar, ntheta;
  int nsteps, warm_steps;           // closed-loop steps done by this launch; warm_steps: step k+1 starts from the solution of step k
  size_t w_step, u_step, x_step, cost_step;   // element offsets per step into plant.w / plant.u_out / plant.x_out / fin.cost
  TubeParams tube;
  TzCsr qmap, hmap, parmap;
  const double* par_lo; const double* par_hi;
  FinishParams fin;
  PlantParams plant;
};

#include "tz_ipm.hip.h"
#include "tz_identify.hip.h"
#include "tz_genstack.hip.h"
#include "tz_gain.hip.h"

// ------------------------------------------------------------------------------------------------
// Finish: one wave per trajectory.
// ------------------------------------------------------------------------------------------------
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
    p.cost[(size_t)b * p.cost_stride] = (st == 0) ? acc / p.cost_scale + r : INFINITY;
  }
  double* v = p.v + (size_t)b * nv;
  for (int c = lane; c < nv; c += 64) {
    if (p.recy) {
      double a = p.rec0[c];
      for (int j = 0; j < n; ++j) a += p.recx[(size_t)c * n + j] * x0[j];
      for (int k = 0; k < nz; ++k) a += p.recy[(size_t)c * nz + k] * x[k];
      v[c] = a;
    } else {
      v[c] = p.Dz[c] * x[p.vpos ? p.vpos[c] : c];
    }
  }
  __syncthreads();                                       // one wave: orders the stores of v above before the loads below
  double* xb = p.xbar + (size_t)b * (N + 1) * n;
  for (int r = lane; r < (N + 1) * n; r += 64) {
    double a = 0.0;
    for (int j = 0; j < n; ++j) a += p.Phi[(size_t)r * n + j] * x0[j];
    const double* g = p.Gam + (size_t)r * nv;
    for (int c = 0; c < nv; ++c) a += g[c] * v[c];
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
