// K0 -- data-driven identification on the device, batched over data sets (Monte Carlo over data seeds).
//
// Reference: TZDDPC.build_zonotopes (tzddpc/tzddpc.py:67-85) and build_zonotopes_theta (:119-128):
//     Xm = x[:-1], Xp = x[1:], Um = u[:-1]                      (:60-62)
//     Mw    = concatenate_zonotope(W, T-1)                      (:81)   centre [c_W ... c_W], one generator per (g_i, column t)
//     Mdata = (Xp' - Mw) pinv([Xm'; Um'])                       (:83)   compute_LTI_matrix_zonotope
//     MdataK = Mdata [I; K],  Mdelta = Mdata - centre           (:119-123)
//     reduce(1) on all three                                    (:126-128)  Girard order 1 = interval box of the generators
// With D = [Xm'; Um'] ((n+m) x (T-1), full row rank) pinv(D) = D'(D D')^-1 =: P, so
//     centre  C = (Xp' - c_W 1') D' (D D')^-1                                       n x (n+m)
//     generators  -g_i (x) P[t, :]   =>  boxed magnitudes  rad(W) s',  s_c  = sum_t |P[t, c]|         (Mdata, Mdelta)
//                                         and              rad(W) sK', sK_c = sum_t |(P [I; K])[t, c]| (MdataK)
// The Gram contractions  [D; Xp' - c_W] D'  run on v_mfma_f64_4x4x4 (the "Hankel-data Gram" of the north star): the four blocks
// of the instruction take four different chunks of four time samples of the SAME output tile, every operand is loaded once and
// serves as the A operand of its tile row and the B operand of its tile column; partial sums are folded with DPP row rotations.
// One wave per data set (the whole problem is (n + m + n) x (T - 1) <= 20 x 399 numbers).
#pragma once

#define TZ_ID_NMAX 8              // K0: dim_x <= 8, dim_u <= 4 (tile counts below)
#define TZ_ID_MMAX 4
#define TZ_ID_PMAX 12            // n + m <= 12
#define TZ_ID_RT 5               // tile rows of Y = [D; Xp' - c_W]: ceil((12 + 8) / 4)
#define TZ_ID_CT 3               // tile columns (rows of D)

struct IdentifyParams {
  int B, T, n, m;
  const double* u;       // B x T x m
  const double* x;       // B x T x n
  const double* wc;      // n          centre of W
  const double* K;       // B x m x n (or m x n when k_shared), may be null: sK, CK are then not produced
  int k_shared;
  double* C;             // B x n x (n+m)      centre of Mdata: [A_hat | B_hat]
  double* s;             // B x (n+m)          sum_t |P[t, :]|
  double* sK;            // B x n              sum_t |(P [I; K])[t, :]|        (may be null)
  double* CK;            // B x n x n          C [I; K]                         (may be null)
  int* status;           // B: 0 ok, 2 the Gram matrix is not positive definite (data not persistently exciting)
};

__global__ __launch_bounds__(64) void tz_identify_kernel(IdentifyParams q) {
  __shared__ double S[TZ_ID_PMAX * TZ_ID_PMAX];          // D D', then its Cholesky factor (lower)
  __shared__ double R[TZ_ID_NMAX * TZ_ID_PMAX];             // (Xp' - c_W) D'
  __shared__ double Si[TZ_ID_PMAX * TZ_ID_PMAX];         // (D D')^-1
  __shared__ int bad;
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = q.n, m = q.m, p = n + m, T1 = q.T - 1;
  const double* xb = q.x + (size_t)b * q.T * n;
  const double* ub = q.u + (size_t)b * q.T * m;
  if (lane == 0) bad = 0;
  // ---- Gram: Y D' with Y = [Xm'; Um'; Xp' - c_W] -----------------------------------------------------------------------
  // operand layout of v_mfma_f64_4x4x4: A[i][k] at lane 16 k + 4 blk + i, B[k][j] at lane 16 k + 4 blk + j, D[i][j] at 16 i + 4 blk + j
  const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
  auto yval = [&](int r, int t) -> double {             // row r of Y at time sample t (0 beyond the data)
    if (t >= T1) return 0.0;
    if (r < n) return xb[(size_t)t * n + r];
    if (r < p) return ub[(size_t)t * m + (r - n)];
    if (r < p + n) return xb[(size_t)(t + 1) * n + (r - p)] - q.wc[r - p];
    return 0.0;
  };
  double acc[TZ_ID_RT][TZ_ID_CT];
#pragma unroll
  for (int a = 0; a < TZ_ID_RT; ++a)
#pragma unroll
    for (int c = 0; c < TZ_ID_CT; ++c) acc[a][c] = 0.0;
  for (int t0 = 0; t0 < T1; t0 += 16) {
    const int t = t0 + 4 * blk + k;
    double v[TZ_ID_RT];
#pragma unroll
    for (int a = 0; a < TZ_ID_RT; ++a) v[a] = yval(4 * a + ij, t);
#pragma unroll
    for (int a = 0; a < TZ_ID_RT; ++a)
#pragma unroll
      for (int c = 0; c < TZ_ID_CT; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[a], v[c], acc[a][c], 0, 0, 0);
  }
  {
    const int i = lane >> 4, j = lane & 3;
#pragma unroll
    for (int a = 0; a < TZ_ID_RT; ++a)
#pragma unroll
      for (int c = 0; c < TZ_ID_CT; ++c) {
        double v = acc[a][c]; v += tz_row_ror<4>(v); v += tz_row_ror<8>(v);      // fold the four time chunks (blk)
        const int r = 4 * a + i, cc = 4 * c + j;
        if (blk == 0 && cc < p) {
          if (r < p) S[r * TZ_ID_PMAX + cc] = v;
          else if (r < p + n) R[(r - p) * TZ_ID_PMAX + cc] = v;
        }
      }
  }
  __syncthreads();
  // ---- Cholesky of S (p <= 12): column by column, lane = row ------------------------------------------------------------
  for (int j = 0; j < p; ++j) {
    if (lane == j) {
      double d = S[j * TZ_ID_PMAX + j];
      for (int kk = 0; kk < j; ++kk) d -= S[j * TZ_ID_PMAX + kk] * S[j * TZ_ID_PMAX + kk];
      if (!(d > 0.0)) { bad = 1; d = 1.0; }
      S[j * TZ_ID_PMAX + j] = sqrt(d);
    }
    __syncthreads();
    if (lane > j && lane < p) {
      double a = S[lane * TZ_ID_PMAX + j];
      for (int kk = 0; kk < j; ++kk) a -= S[lane * TZ_ID_PMAX + kk] * S[j * TZ_ID_PMAX + kk];
      S[lane * TZ_ID_PMAX + j] = a / S[j * TZ_ID_PMAX + j];
    }
    __syncthreads();
  }
  // ---- (D D')^-1: lane c < p solves L L' z = e_c --------------------------------------------------------------------------
  if (lane < p) {
    double z[TZ_ID_PMAX];
#pragma unroll
    for (int i = 0; i < TZ_ID_PMAX; ++i) z[i] = (i == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < TZ_ID_PMAX; ++i) {
      if (i < p) {
        double a = z[i];
#pragma unroll
        for (int kk = 0; kk < TZ_ID_PMAX; ++kk) if (kk < i) a -= S[i * TZ_ID_PMAX + kk] * z[kk];
        z[i] = a / S[i * TZ_ID_PMAX + i];
      }
    }
#pragma unroll
    for (int ii = 0; ii < TZ_ID_PMAX; ++ii) {
      const int i = TZ_ID_PMAX - 1 - ii;
      if (i < p) {
        double a = z[i];
#pragma unroll
        for (int kk = 0; kk < TZ_ID_PMAX; ++kk) if (kk > i && kk < p) a -= S[kk * TZ_ID_PMAX + i] * z[kk];
        z[i] = a / S[i * TZ_ID_PMAX + i];
      }
    }
#pragma unroll
    for (int i = 0; i < TZ_ID_PMAX; ++i) if (i < p) Si[i * TZ_ID_PMAX + lane] = z[i];
  }
  __syncthreads();
  // ---- centre C = R (D D')^-1 and C [I; K] -------------------------------------------------------------------------------
  const double* Kb = q.K ? q.K + (q.k_shared ? 0 : (size_t)b * m * n) : nullptr;
  double* Cb = q.C + (size_t)b * n * p;
  for (int e = lane; e < n * p; e += 64) {
    const int r = e / p, c = e - r * p;
    double a = 0.0;
    for (int kk = 0; kk < p; ++kk) a += R[r * TZ_ID_PMAX + kk] * Si[kk * TZ_ID_PMAX + c];
    Cb[e] = a;
  }
  __syncthreads();
  if (Kb && q.CK) {
    for (int e = lane; e < n * n; e += 64) {
      const int r = e / n, c = e - r * n;
      double a = Cb[r * p + c];
      for (int j = 0; j < m; ++j) a += Cb[r * p + n + j] * Kb[j * n + c];
      q.CK[(size_t)b * n * n + e] = a;
    }
  }
  // ---- s = sum_t |P[t, :]|, sK = sum_t |P[t, :] [I; K]|  with P[t, :] = (D D')^-1 D[:, t]; lanes over t ------------------------
  double ss[TZ_ID_PMAX], sk[TZ_ID_NMAX];
#pragma unroll
  for (int c = 0; c < TZ_ID_PMAX; ++c) ss[c] = 0.0;
#pragma unroll
  for (int c = 0; c < TZ_ID_NMAX; ++c) sk[c] = 0.0;
  for (int t = lane; t < T1; t += 64) {
    double d[TZ_ID_PMAX], pr[TZ_ID_PMAX];
#pragma unroll
    for (int r = 0; r < TZ_ID_PMAX; ++r) d[r] = (r < p) ? yval(r, t) : 0.0;
#pragma unroll
    for (int c = 0; c < TZ_ID_PMAX; ++c) {
      double a = 0.0;
      if (c < p) {
#pragma unroll
        for (int r = 0; r < TZ_ID_PMAX; ++r) if (r < p) a += Si[c * TZ_ID_PMAX + r] * d[r];
      }
      pr[c] = a; ss[c] += fabs(a);
    }
    if (Kb) {
#pragma unroll
      for (int c = 0; c < TZ_ID_NMAX; ++c) {
        if (c < n) {
          double a = pr[c];
#pragma unroll
          for (int c2 = 0; c2 < TZ_ID_PMAX; ++c2) if (c2 >= n && c2 < p) a += pr[c2] * Kb[(c2 - n) * n + c];
          sk[c] += fabs(a);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < TZ_ID_PMAX; ++c) { const double v = tz_wave_reduce<RED_SUM>(ss[c]); if (lane == 0 && c < p) q.s[(size_t)b * p + c] = v; }
  if (Kb && q.sK) {
#pragma unroll
    for (int c = 0; c < TZ_ID_NMAX; ++c) { const double v = tz_wave_reduce<RED_SUM>(sk[c]); if (lane == 0 && c < n) q.sK[(size_t)b * n + c] = v; }
  }
  if (lane == 0) q.status[b] = bad ? 2 : 0;
}
