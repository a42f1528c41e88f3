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
  double* ws;            // B x (pmax+1) x (3n + m): [c | beta | radx | radu] per power
};

__global__ void tz_tube_kernel(TubeParams p) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  const int n = p.n, m = p.m;
  const int stride = 3 * n + m;
  double* ws = p.ws + (size_t)b * (p.pmax + 1) * stride;
  double* th = p.theta + (size_t)b * p.ntheta;
  for (int i = 0; i < n; ++i) {
    double x0 = p.xbar0[(size_t)b * n + i];
    th[i] = x0; th[n + i] = fabs(x0);
    ws[i] = p.e0[(size_t)b * n + i];          // c^(0)
    ws[2 * n + i] = 0.0;                      // radx^(0)
  }
  for (int j = 0; j < m; ++j) ws[3 * n + j] = 0.0;
  for (int pw = 0; pw < p.pmax; ++pw) {
    double* cur = ws + (size_t)pw * stride;
    double* nxt = cur + stride;
    // beta_pw = DK (|c| + radx)
    for (int i = 0; i < n; ++i) {
      double acc = 0.0;
      for (int j = 0; j < n; ++j) acc += p.DK[i * n + j] * (fabs(cur[j]) + cur[2 * n + j]);
      cur[n + i] = acc;
    }
    for (int i = 0; i < n; ++i) {
      double acc = 0.0;
      for (int j = 0; j < n; ++j) acc += p.CK[i * n + j] * cur[j];
      nxt[i] = acc;
    }
    for (int i = 0; i < n; ++i) {
      double acc = 0.0;
      for (int l = 0; l <= pw; ++l) {
        const double* M = p.absCK + (size_t)(pw - l) * n * n + i * n;
        const double* beta = ws + (size_t)l * stride + n;
        for (int j = 0; j < n; ++j) acc += M[j] * beta[j];
      }
      nxt[2 * n + i] = acc;
    }
    for (int i = 0; i < m; ++i) {
      double acc = 0.0;
      for (int l = 0; l <= pw; ++l) {
        const double* M = p.absKCK + (size_t)(pw - l) * m * n + i * n;
        const double* beta = ws + (size_t)l * stride + n;
        for (int j = 0; j < n; ++j) acc += M[j] * beta[j];
      }
      nxt[3 * n + i] = acc;
    }
  }
  const int blk = 2 * n + m;
  for (int k = 0; k < p.N; ++k) {
    const double* src = ws + (size_t)p.power[k] * stride;
    double* dst = th + 2 * n + k * blk;
    for (int i = 0; i < n; ++i) { dst[i] = src[i]; dst[n + i] = src[2 * n + i]; }
    for (int j = 0; j < m; ++j) dst[2 * n + j] = src[3 * n + j];
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

// ------------------------------------------------------------------------------------------------
// Interior point kernel: one 256-thread workgroup per trajectory.
// ------------------------------------------------------------------------------------------------
struct IpmItem { int I0, q0, nq, kptr, klen; };

struct IpmParams {
  int B, nz, mi, nzp, mip, Tz, Kc, nquads;
  const double* P;       // nzp x nzp
  const double* G;       // mip x nzp   (row-major, zero padded)
  const double* Gt;      // nzp x mip   (transpose)
  const double* Gp;      // Kc x Tz x 16 patches: Gp[(kc*Tz + J)*16 + 4k + j] = G[4kc+k][4J+j]
  const IpmItem* items;  // Gram work items, grouped per wave
  const int* item_ptr;   // TZ_NWAVES + 1
  const int* klist;
  const double* q; const double* h;
  const int* prestatus;
  double* x; double* s; double* lam;
  int* status; int* iters;
  int max_iter; double tol, reg, step_frac;
};

__device__ inline int tz_qprefix(int I) {   // number of quads in tile rows < I (row I has (I>>2)+1 quads)
  int a = I >> 2, b = I & 3;
  return (a + 1) * (2 * a + b);
}
__device__ inline int tz_hidx(int r, int c) {   // LDS index of H(r, c), r >= c (tile-row major, quads in lane order)
  int I = r >> 2, J = c >> 2;
  return (tz_qprefix(I) + (J >> 2)) * 64 + 16 * (r & 3) + 4 * (J & 3) + (c & 3);
}

enum { RED_SUM = 0, RED_MAX = 1, RED_MIN = 2 };

template <int OP>
__device__ inline double tz_wave_reduce(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o = __shfl_down(v, off, 64);
    if (OP == RED_SUM) v += o;
    else if (OP == RED_MAX) v = fmax(v, o);
    else v = fmin(v, o);
  }
  return v;
}

// three simultaneous block reductions (ops fixed at compile time); result broadcast to all threads.
template <int OP0, int OP1, int OP2>
__device__ inline void tz_block_reduce3(double& a, double& b, double& c, double* red) {
  a = tz_wave_reduce<OP0>(a); b = tz_wave_reduce<OP1>(b); c = tz_wave_reduce<OP2>(c);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();                       // protect red[] from the previous use
  if (lane == 0) { red[w] = a; red[4 + w] = b; red[8 + w] = c; }
  __syncthreads();
  double ra = red[0], rb = red[4], rc = red[8];
#pragma unroll
  for (int i = 1; i < TZ_NWAVES; ++i) {
    double va = red[i], vb = red[4 + i], vc = red[8 + i];
    ra = (OP0 == RED_SUM) ? ra + va : (OP0 == RED_MAX ? fmax(ra, va) : fmin(ra, va));
    rb = (OP1 == RED_SUM) ? rb + vb : (OP1 == RED_MAX ? fmax(rb, vb) : fmin(rb, vb));
    rc = (OP2 == RED_SUM) ? rc + vc : (OP2 == RED_MAX ? fmax(rc, vc) : fmin(rc, vc));
  }
  a = ra; b = rb; c = rc;
}

// out[r] = sum_c Gt[c][r] in[c]   (r < mi); `in` in LDS
__device__ inline void tz_gemv_G(const IpmParams& p, double* out, const double* in) {
  for (int r = threadIdx.x; r < p.mi; r += TZ_THREADS) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const double* g = p.Gt + r;
    int c = 0;
    for (; c + 3 < p.nz; c += 4) {
      a0 += g[(size_t)c * p.mip] * in[c];
      a1 += g[(size_t)(c + 1) * p.mip] * in[c + 1];
      a2 += g[(size_t)(c + 2) * p.mip] * in[c + 2];
      a3 += g[(size_t)(c + 3) * p.mip] * in[c + 3];
    }
    for (; c < p.nz; ++c) a0 += g[(size_t)c * p.mip] * in[c];
    out[r] = (a0 + a1) + (a2 + a3);
  }
}

// out[c] (+)= sum_r M[r][c] in[r]  for a row-major matrix M (rows x nzp);  part: TZ_NWAVES x nzp scratch.
// Ends with the partial sums in part[]; caller combines after a barrier via tz_gemvT_combine.
__device__ inline void tz_gemvT_partial(const double* M, int rows, int nzp, const double* in, double* part) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double acc[4] = {0, 0, 0, 0};        // columns lane, lane+64, lane+128, lane+192  (nzp <= 256)
  for (int r = w; r < rows; r += TZ_NWAVES) {
    double v = in[r];
    const double* row = M + (size_t)r * nzp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int c = lane + 64 * g;
      if (c < nzp) acc[g] += row[c] * v;
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    int c = lane + 64 * g;
    if (c < nzp) part[w * nzp + c] = acc[g];
  }
}
__device__ inline double tz_gemvT_get(const double* part, int nzp, int c) {
  return (part[c] + part[nzp + c]) + (part[2 * nzp + c] + part[3 * nzp + c]);
}

// Gram matrix  H = P + G' diag(w) G + reg I  into LDS quads, by v_mfma_f64_4x4x4 (blk = 4 column tiles).
__device__ inline void tz_form_H(const IpmParams& p, double* Hq, const double* wv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
  const int Tz = p.Tz;
  for (int it = p.item_ptr[wave]; it < p.item_ptr[wave + 1]; ++it) {
    const IpmItem item = p.items[it];
    double acc[4][2];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) { acc[ii][0] = 0.0; acc[ii][1] = 0.0; }
    int Jb[2]; bool Jok[2];
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) { Jb[nn] = 4 * (item.q0 + nn) + blk; Jok[nn] = (nn < item.nq) && (Jb[nn] < Tz); }
    for (int kk = 0; kk < item.klen; ++kk) {
      const int kc = p.klist[item.kptr + kk];
      const double wk = wv[4 * kc + k];
      const double* prow = p.Gp + (size_t)kc * Tz * 16 + 4 * k + ij;
      double a[4], bq[2];
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        int I = item.I0 + ii;
        a[ii] = (I < Tz) ? prow[I * 16] * wk : 0.0;
      }
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) bq[nn] = Jok[nn] ? prow[Jb[nn] * 16] : 0.0;
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        acc[ii][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[ii], bq[0], acc[ii][0], 0, 0, 0);
        acc[ii][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[ii], bq[1], acc[ii][1], 0, 0, 0);
      }
    }
    // D lane (i = lane>>4, blk, j = lane&3) = H(4I + i, 4(4q + blk) + j)
    const int i = lane >> 4, j = lane & 3;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      int I = item.I0 + ii;
      if (I >= Tz) continue;
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        int q = item.q0 + nn;
        if (nn >= item.nq || q > (I >> 2)) continue;
        int r = 4 * I + i, c = 4 * (4 * q + blk) + j;
        double v = acc[ii][nn];
        if (c < p.nzp) v += p.P[(size_t)r * p.nzp + c];
        if (r == c) v = (r < p.nz) ? v + p.reg : 1.0;
        Hq[(tz_qprefix(I) + q) * 64 + lane] = v;
      }
    }
  }
}

// In-place blocked Cholesky (tile 4) of the quad-stored matrix; dinv[p] = inverse of the diagonal tile's factor.
// Returns (uniformly) false if a pivot was not positive.
__device__ inline bool tz_cholesky(const IpmParams& p, double* Hq, double* dinv, int* flag) {
  const int Tz = p.Tz;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int pp = 0; pp < Tz; ++pp) {
    const int dbase = (tz_qprefix(pp) + (pp >> 2)) * 64 + 4 * (pp & 3);
    const double a00 = Hq[dbase], a10 = Hq[dbase + 16], a11 = Hq[dbase + 17];
    const double a20 = Hq[dbase + 32], a21 = Hq[dbase + 33], a22 = Hq[dbase + 34];
    const double a30 = Hq[dbase + 48], a31 = Hq[dbase + 49], a32 = Hq[dbase + 50], a33 = Hq[dbase + 51];
    bool ok = true;
    double d0 = a00; ok = ok && (d0 > 0.0);
    const double l00 = sqrt(fmax(d0, 1e-300)), i00 = 1.0 / l00;
    const double l10 = a10 * i00, l20 = a20 * i00, l30 = a30 * i00;
    double d1 = a11 - l10 * l10; ok = ok && (d1 > 0.0);
    const double l11 = sqrt(fmax(d1, 1e-300)), i11 = 1.0 / l11;
    const double l21 = (a21 - l20 * l10) * i11, l31 = (a31 - l30 * l10) * i11;
    double d2 = a22 - l20 * l20 - l21 * l21; ok = ok && (d2 > 0.0);
    const double l22 = sqrt(fmax(d2, 1e-300)), i22 = 1.0 / l22;
    const double l32 = (a32 - l30 * l20 - l31 * l21) * i22;
    double d3 = a33 - l30 * l30 - l31 * l31 - l32 * l32; ok = ok && (d3 > 0.0);
    const double l33 = sqrt(fmax(d3, 1e-300)), i33 = 1.0 / l33;
    __syncthreads();   // every thread has read the diagonal tile before anyone overwrites it
    if (threadIdx.x == 0) {
      if (!ok) *flag = 1;
      Hq[dbase] = l00; Hq[dbase + 1] = 0; Hq[dbase + 2] = 0; Hq[dbase + 3] = 0;
      Hq[dbase + 16] = l10; Hq[dbase + 17] = l11; Hq[dbase + 18] = 0; Hq[dbase + 19] = 0;
      Hq[dbase + 32] = l20; Hq[dbase + 33] = l21; Hq[dbase + 34] = l22; Hq[dbase + 35] = 0;
      Hq[dbase + 48] = l30; Hq[dbase + 49] = l31; Hq[dbase + 50] = l32; Hq[dbase + 51] = l33;
      const double m10 = -l10 * i00 * i11;
      const double m21 = -l21 * i11 * i22;
      const double m32 = -l32 * i22 * i33;
      const double m20 = -(l20 * i00 + l21 * m10) * i22;
      const double m31 = -(l31 * i11 + l32 * m21) * i33;
      const double m30 = -(l30 * i00 + l31 * m10 + l32 * m20) * i33;
      double* di = dinv + pp * 16;
      di[0] = i00; di[1] = 0; di[2] = 0; di[3] = 0;
      di[4] = m10; di[5] = i11; di[6] = 0; di[7] = 0;
      di[8] = m20; di[9] = m21; di[10] = i22; di[11] = 0;
      di[12] = m30; di[13] = m31; di[14] = m32; di[15] = i33;
    }
    // panel: rows of tiles (I, pp), I > pp:  x L_pp' = a
    for (int t = threadIdx.x; t < 4 * (Tz - pp - 1); t += TZ_THREADS) {
      const int I = pp + 1 + (t >> 2), i = t & 3;
      const int base = (tz_qprefix(I) + (pp >> 2)) * 64 + 16 * i + 4 * (pp & 3);
      const double b0 = Hq[base], b1 = Hq[base + 1], b2 = Hq[base + 2], b3 = Hq[base + 3];
      const double x0 = b0 * i00;
      const double x1 = (b1 - x0 * l10) * i11;
      const double x2 = (b2 - x0 * l20 - x1 * l21) * i22;
      const double x3 = (b3 - x0 * l30 - x1 * l31 - x2 * l32) * i33;
      Hq[base] = x0; Hq[base + 1] = x1; Hq[base + 2] = x2; Hq[base + 3] = x3;
    }
    __syncthreads();
    // trailing update  H(I, J) -= L(I, pp) L(J, pp)'  for pp < J <= I, four column tiles per MFMA
    {
      const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
      int e = 0;
      for (int I = pp + 1; I < Tz; ++I) {
        const int qI = tz_qprefix(I);
        for (int q = (pp + 1) >> 2; q <= (I >> 2); ++q, ++e) {
          if ((e & (TZ_NWAVES - 1)) != wave) continue;
          const double a = -Hq[(qI + (pp >> 2)) * 64 + 16 * ij + 4 * (pp & 3) + k];     // -L(4I+i, 4pp+k), i = ij
          const int J = 4 * q + blk;
          const bool valid = (J > pp) && (J <= I);
          const double bb = valid ? Hq[(tz_qprefix(J) + (pp >> 2)) * 64 + 16 * ij + 4 * (pp & 3) + k] : 0.0;  // L(4J+j, 4pp+k), j = ij
          const int ci = (qI + q) * 64 + lane;
          const double c = Hq[ci];
          const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, c, 0, 0, 0);
          if (valid) Hq[ci] = d;
        }
      }
    }
    __syncthreads();
  }
  return *flag == 0;
}

// Solve (L L') out = rhs.  rhs is destroyed; tmp receives the forward solution.  All in LDS, nzp entries.
__device__ inline void tz_chol_solve(const IpmParams& p, const double* Hq, const double* dinv,
                                     double* rhs, double* tmp, double* out) {
  const int Tz = p.Tz, nzp = p.nzp;
  const int t = threadIdx.x;
  // forward: L y = rhs
  for (int I = 0; I < Tz; ++I) {
    const double r0 = rhs[4 * I], r1 = rhs[4 * I + 1], r2 = rhs[4 * I + 2], r3 = rhs[4 * I + 3];
    const double* di = dinv + I * 16;
    const double y0 = di[0] * r0;
    const double y1 = di[4] * r0 + di[5] * r1;
    const double y2 = di[8] * r0 + di[9] * r1 + di[10] * r2;
    const double y3 = di[12] * r0 + di[13] * r1 + di[14] * r2 + di[15] * r3;
    if (t == 0) { tmp[4 * I] = y0; tmp[4 * I + 1] = y1; tmp[4 * I + 2] = y2; tmp[4 * I + 3] = y3; }
    for (int r = 4 * (I + 1) + t; r < nzp; r += TZ_THREADS) {
      const int base = tz_hidx(r, 4 * I);
      rhs[r] -= Hq[base] * y0 + Hq[base + 1] * y1 + Hq[base + 2] * y2 + Hq[base + 3] * y3;
    }
    __syncthreads();
  }
  // backward: L' out = tmp
  for (int I = Tz - 1; I >= 0; --I) {
    const double y0 = tmp[4 * I], y1 = tmp[4 * I + 1], y2 = tmp[4 * I + 2], y3 = tmp[4 * I + 3];
    const double* di = dinv + I * 16;     // x = M' y  (M lower)
    const double x3 = di[15] * y3;
    const double x2 = di[10] * y2 + di[14] * y3;
    const double x1 = di[5] * y1 + di[9] * y2 + di[13] * y3;
    const double x0 = di[0] * y0 + di[4] * y1 + di[8] * y2 + di[12] * y3;
    __syncthreads();     // all threads have read tmp[4I..] before the updates below touch lower entries
    if (t == 0) { out[4 * I] = x0; out[4 * I + 1] = x1; out[4 * I + 2] = x2; out[4 * I + 3] = x3; }
    for (int c = t; c < 4 * I; c += TZ_THREADS) {
      const int base = (tz_qprefix(I) + (c >> 4)) * 64 + 4 * ((c >> 2) & 3) + (c & 3);
      tmp[c] -= Hq[base] * x0 + Hq[base + 16] * x1 + Hq[base + 32] * x2 + Hq[base + 48] * x3;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(TZ_THREADS) void tz_ipm_kernel(IpmParams p) {
  extern __shared__ double lds[];
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int nz = p.nz, mi = p.mi, nzp = p.nzp, mip = p.mip;
  if (p.prestatus[b] != 0) {
    if (t == 0) { p.status[b] = 3; p.iters[b] = 0; }
    for (int c = t; c < nz; c += TZ_THREADS) p.x[(size_t)b * nz + c] = 0.0;
    for (int r = t; r < mi; r += TZ_THREADS) { p.s[(size_t)b * mi + r] = 1.0; p.lam[(size_t)b * mi + r] = 0.0; }
    return;
  }
  double* Hq = lds;
  double* dinv = Hq + (size_t)p.nquads * 64;
  double* xv = dinv + p.Tz * 16;
  double* dxv = xv + nzp;
  double* rdv = dxv + nzp;
  double* r1v = rdv + nzp;
  double* qv = r1v + nzp;
  double* tmpz = qv + nzp;
  double* part = tmpz + nzp;              // 4 * nzp
  double* sv = part + 4 * nzp;
  double* lv = sv + mip;
  double* dsv = lv + mip;
  double* dlv = dsv + mip;
  double* wv = dlv + mip;
  double* rpv = wv + mip;
  double* hv = rpv + mip;
  double* gxv = hv + mip;
  double* t1v = gxv + mip;
  double* gdx = t1v + mip;
  double* red = gdx + mip;                // 16
  int* flag = (int*)(red + 16);

  for (int c = t; c < nzp; c += TZ_THREADS) { qv[c] = (c < nz) ? p.q[(size_t)b * nz + c] : 0.0; xv[c] = 0.0; }
  for (int r = t; r < mip; r += TZ_THREADS) {
    hv[r] = (r < mi) ? p.h[(size_t)b * mi + r] : 0.0;
    wv[r] = (r < mi) ? 1.0 : 0.0;
    sv[r] = 1.0; lv[r] = (r < mi) ? 1.0 : 0.0;
  }
  if (t == 0) *flag = 0;
  __syncthreads();

  // ---- start point: (P + G'G + reg) x = -q + G'h, then shift the slacks into the cone ----------
  tz_form_H(p, Hq, wv);
  tz_gemvT_partial(p.G, mi, nzp, hv, part);
  __syncthreads();
  for (int c = t; c < nzp; c += TZ_THREADS) r1v[c] = (c < nz) ? tz_gemvT_get(part, nzp, c) - qv[c] : 0.0;
  __syncthreads();
  bool okf = tz_cholesky(p, Hq, dinv, flag);
  tz_chol_solve(p, Hq, dinv, r1v, tmpz, xv);
  __syncthreads();
  tz_gemv_G(p, gxv, xv);
  __syncthreads();
  {
    double rmin = 1e300, d1 = 0, d2 = 0;
    for (int r = t; r < mi; r += TZ_THREADS) rmin = fmin(rmin, hv[r] - gxv[r]);
    tz_block_reduce3<RED_MIN, RED_SUM, RED_SUM>(rmin, d1, d2, red);
    const double shift = (rmin <= 1e-8) ? fmax(0.0, 1.0 - rmin) : 0.0;
    for (int r = t; r < mi; r += TZ_THREADS) sv[r] = hv[r] - gxv[r] + shift;
  }
  double scq = 0, sch = 0, dummy = 0;
  for (int c = t; c < nz; c += TZ_THREADS) scq = fmax(scq, fabs(qv[c]));
  for (int r = t; r < mi; r += TZ_THREADS) sch = fmax(sch, fabs(hv[r]));
  tz_block_reduce3<RED_MAX, RED_MAX, RED_SUM>(scq, sch, dummy, red);
  const double sc_d = 1.0 + scq, sc_p = 1.0 + sch;

  int status = 1, it = 0;
  if (!okf) status = 2;
  for (it = 0; it < p.max_iter && status == 1; ++it) {
    // residuals: rd = P x + q + G'lam ; rp = G x + s - h ; mu
    tz_gemvT_partial(p.G, mi, nzp, lv, part);
    __syncthreads();
    for (int c = t; c < nzp; c += TZ_THREADS) rdv[c] = (c < nz) ? tz_gemvT_get(part, nzp, c) + qv[c] : 0.0;
    __syncthreads();
    tz_gemvT_partial(p.P, nz, nzp, xv, part);
    __syncthreads();
    double nrd = 0, nrp = 0, sl = 0;
    for (int c = t; c < nz; c += TZ_THREADS) { double v = rdv[c] + tz_gemvT_get(part, nzp, c); rdv[c] = v; nrd = fmax(nrd, fabs(v)); }
    for (int r = t; r < mi; r += TZ_THREADS) {
      double v = gxv[r] + sv[r] - hv[r]; rpv[r] = v; nrp = fmax(nrp, fabs(v));
      sl += sv[r] * lv[r];
    }
    tz_block_reduce3<RED_MAX, RED_MAX, RED_SUM>(nrd, nrp, sl, red);
    const double mu = sl / mi;
    nrd /= sc_d; nrp /= sc_p;
    if (nrd <= p.tol && nrp <= p.tol && mu <= p.tol) { status = 0; break; }
    if (mu <= 1e-3 * p.tol) { status = (nrd <= 1e3 * p.tol && nrp <= 1e3 * p.tol) ? 0 : 3; break; }
    if (!(mu == mu) || !(nrd == nrd) || mu > 1e200) { status = 2; break; }
    // Newton matrix
    for (int r = t; r < mi; r += TZ_THREADS) wv[r] = lv[r] / sv[r];
    __syncthreads();
    tz_form_H(p, Hq, wv);
    __syncthreads();
    if (!tz_cholesky(p, Hq, dinv, flag)) { status = 2; break; }
    // ---- predictor: rc = s*lam ------------------------------------------------------------------
    for (int r = t; r < mi; r += TZ_THREADS) t1v[r] = wv[r] * rpv[r] - lv[r];      // (-rc + lam rp)/s
    __syncthreads();
    tz_gemvT_partial(p.G, mi, nzp, t1v, part);
    __syncthreads();
    for (int c = t; c < nzp; c += TZ_THREADS) r1v[c] = (c < nz) ? -rdv[c] - tz_gemvT_get(part, nzp, c) : 0.0;
    __syncthreads();
    tz_chol_solve(p, Hq, dinv, r1v, tmpz, dxv);
    __syncthreads();
    tz_gemv_G(p, gdx, dxv);
    __syncthreads();
    double ap = 1.0, ad = 1.0, z0 = 0;
    for (int r = t; r < mi; r += TZ_THREADS) {
      const double ds = -rpv[r] - gdx[r];
      const double dl = -lv[r] - wv[r] * ds;
      dsv[r] = ds; dlv[r] = dl;
      if (ds < 0) ap = fmin(ap, -sv[r] / ds);
      if (dl < 0) ad = fmin(ad, -lv[r] / dl);
    }
    tz_block_reduce3<RED_MIN, RED_MIN, RED_SUM>(ap, ad, z0, red);
    double muaff = 0, z1 = 0, z2 = 0;
    for (int r = t; r < mi; r += TZ_THREADS) muaff += (sv[r] + ap * dsv[r]) * (lv[r] + ad * dlv[r]);
    tz_block_reduce3<RED_SUM, RED_SUM, RED_SUM>(muaff, z1, z2, red);
    muaff /= mi;
    double sigma = muaff / mu; sigma = sigma * sigma * sigma;
    // ---- corrector: rc = s*lam + dsa*dla - sigma mu -----------------------------------------------
    for (int r = t; r < mi; r += TZ_THREADS) {
      const double rc = sv[r] * lv[r] + dsv[r] * dlv[r] - sigma * mu;
      t1v[r] = (-rc + lv[r] * rpv[r]) / sv[r];
      dsv[r] = rc;                                  // keep rc for the dl formula
    }
    __syncthreads();
    tz_gemvT_partial(p.G, mi, nzp, t1v, part);
    __syncthreads();
    for (int c = t; c < nzp; c += TZ_THREADS) r1v[c] = (c < nz) ? -rdv[c] - tz_gemvT_get(part, nzp, c) : 0.0;
    __syncthreads();
    tz_chol_solve(p, Hq, dinv, r1v, tmpz, dxv);
    __syncthreads();
    tz_gemv_G(p, gdx, dxv);
    __syncthreads();
    double as = 1e300, al = 1e300, z3 = 0;
    for (int r = t; r < mi; r += TZ_THREADS) {
      const double rc = dsv[r];
      const double ds = -rpv[r] - gdx[r];
      const double dl = (-rc - lv[r] * ds) / sv[r];
      dsv[r] = ds; dlv[r] = dl;
      if (ds < 0) as = fmin(as, -sv[r] / ds);
      if (dl < 0) al = fmin(al, -lv[r] / dl);
    }
    tz_block_reduce3<RED_MIN, RED_MIN, RED_SUM>(as, al, z3, red);
    const double alpha = fmin(1.0, p.step_frac * fmin(as, al));
    for (int c = t; c < nz; c += TZ_THREADS) xv[c] += alpha * dxv[c];
    for (int r = t; r < mi; r += TZ_THREADS) {
      sv[r] += alpha * dsv[r]; lv[r] += alpha * dlv[r]; gxv[r] += alpha * gdx[r];
    }
    __syncthreads();
  }
  for (int c = t; c < nz; c += TZ_THREADS) p.x[(size_t)b * nz + c] = xv[c];
  for (int r = t; r < mi; r += TZ_THREADS) { p.s[(size_t)b * mi + r] = sv[r]; p.lam[(size_t)b * mi + r] = lv[r]; }
  if (t == 0) { p.status[b] = status; p.iters[b] = it; }
}

// ------------------------------------------------------------------------------------------------
// Finish: one wave per trajectory.
// ------------------------------------------------------------------------------------------------
struct FinishParams {
  int B, n, m, N, nz, mi, nzp, nc_rows;
  const double* P;  const double* Dz; const double* Phi; const double* Gam;
  const double* r1; const double* R2; double r0; double cost_scale;
  const int* row_of;
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
      if (s[r] < lam[r]) act[p.row_of[r]] = 1;
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
