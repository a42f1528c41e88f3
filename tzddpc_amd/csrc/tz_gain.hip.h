// Gain synthesis on the device (reference tzddpc/utils.py:13-41, :105-129), batched over samples.
//
// Both pieces work on closed-loop matrices drawn from a matrix zonotope,   M(beta) = M0 + sum_i beta_i H_i   (n x n, n <= 8):
//   * is_gain_robust (:105-129): H_i = G_i[:, :n] + G_i[:, n:] K, beta uniform in [-1, 1]^gamma (Mdata.sample()), the test is
//     spectral_radius(M) < 1 for every one of ceil(log(1/conf) / log(1/(1-acc))) samples (1146 at the defaults)
//     -> tz_specrad_kernel: lane = sample, the spectral radius from an in-LDS Hessenberg reduction + Francis double-shift QR.
//   * compute_A_B (:13-41): maximise ||A + B K||_F over INDEPENDENT beta_A, beta_B in [-1, 1]^gamma (the reference scales the A and
//     the B columns of every generator separately, :19-32), i.e. H = [G_i[:, :n]]_i ++ [G_i[:, n:] K]_i.  The reference hands
//     this to DCCP (convex-concave procedure: linearise the convex objective, maximise the linearisation over the box) from
//     num_init random points; over a box the maximiser of the linearisation is the vertex beta_i = sign <M, H_i>, so one CCP step is
//     a sign update of all beta at once -> tz_adversary_kernel: lane = starting point, iterate to the fixed point.
// The generators are streamed global -> LDS in tiles (every lane needs every generator: LDS broadcast reads), the sample's own
// matrix lives in registers (accumulation) and in an LDS slab laid out [entry][lane] (dynamic indexing of the QR sweeps; a lane
// only touches its own column of the slab: bank-conflict free, no barriers inside the QR).
#pragma once

#define TZ_GN_TILE 32            // generators per LDS tile (32 x 64 doubles = 16 KB)
#define TZ_GN_NMAX 8             // n <= 8 (TZ_NMAX)

struct GainParams {
  int S, n, ngen, max_iter;
  const double* M0;              // n x n
  const double* H;               // ngen x n x n
  const double* beta_in;         // S x ngen    (specrad: the samples; adversary: the starting points)
  double* beta_out;              // S x ngen    (adversary only)
  double* val;                   // S           spectral radius / Frobenius norm at the fixed point
  int* aux;                      // S           specrad: 0 ok, 2 the QR iteration did not converge; adversary: CCP steps taken
};

// M (registers, entries >= n*n untouched) = M0 + sum_g beta[g] H_g for this lane's sample; all lanes of the block run the loop
__device__ inline void tz_gain_accumulate(const GainParams& q, double* tile, int s, bool live, double (&M)[TZ_GN_NMAX * TZ_GN_NMAX]) {
  const int n2 = q.n * q.n, t = threadIdx.x;
#pragma unroll
  for (int e = 0; e < TZ_GN_NMAX * TZ_GN_NMAX; ++e) M[e] = (e < n2) ? q.M0[e] : 0.0;
  for (int g0 = 0; g0 < q.ngen; g0 += TZ_GN_TILE) {
    const int ng = min(TZ_GN_TILE, q.ngen - g0);
    __syncthreads();
    for (int i = t; i < ng * n2; i += blockDim.x) tile[i] = q.H[(size_t)g0 * n2 + i];
    __syncthreads();
    // this lane's coefficients of the tile: requested together (a lane's row of beta is contiguous), not one dependent 8-byte load
    // per generator -- round 2's version spent its time in those round trips (profiles/r3j_aux_kernel_stats.csv)
    double bt[TZ_GN_TILE];
#pragma unroll
    for (int g = 0; g < TZ_GN_TILE; ++g) bt[g] = (live && g < ng) ? q.beta_in[(size_t)s * q.ngen + g0 + g] : 0.0;
#pragma unroll
    for (int g = 0; g < TZ_GN_TILE; ++g) {
      if (g < ng) {
        const double* h = tile + g * n2;
#pragma unroll
        for (int e = 0; e < TZ_GN_NMAX * TZ_GN_NMAX; ++e) if (e < n2) M[e] += bt[g] * h[e];
      }
    }
  }
}

// Spectral radius of the n x n matrix in this lane's slab column, a(i, j) = slab[(i * n + j) * 64].  Elementary stabilised
// similarity transformations to Hessenberg form, then the shifted QR iteration with deflation (Francis double shift; the
// classic EISPACK elmhes / hqr pair, eigenvalues only).  Returns false when a block needs more than 60 sweeps.
__device__ inline bool tz_spectral_radius(double* slab, int n, double& rho) {
#define A_(i, j) slab[((i) * n + (j)) << 6]
  // ---- Hessenberg reduction ----------------------------------------------------------------------------------------
  for (int mcol = 1; mcol < n - 1; ++mcol) {
    double x = 0.0; int ip = mcol;
    for (int j = mcol; j < n; ++j) if (fabs(A_(j, mcol - 1)) > fabs(x)) { x = A_(j, mcol - 1); ip = j; }
    if (ip != mcol) {
      for (int j = mcol - 1; j < n; ++j) { const double tmp = A_(ip, j); A_(ip, j) = A_(mcol, j); A_(mcol, j) = tmp; }
      for (int j = 0; j < n; ++j) { const double tmp = A_(j, ip); A_(j, ip) = A_(j, mcol); A_(j, mcol) = tmp; }
    }
    if (x != 0.0) {
      for (int i = mcol + 1; i < n; ++i) {
        double y = A_(i, mcol - 1);
        if (y != 0.0) {
          y /= x;
          for (int j = mcol; j < n; ++j) A_(i, j) -= y * A_(mcol, j);
          for (int j = 0; j < n; ++j) A_(j, mcol) += y * A_(j, i);
        }
      }
    }
  }
  for (int i = 2; i < n; ++i) for (int j = 0; j < i - 1; ++j) A_(i, j) = 0.0;
  // ---- QR iteration ---------------------------------------------------------------------------------------------------
  double anorm = 0.0;
  for (int i = 0; i < n; ++i) for (int j = max(i - 1, 0); j < n; ++j) anorm += fabs(A_(i, j));
  rho = 0.0;
  if (!(anorm < 1e300)) return false;                    // non-finite input
  int nn = n - 1, its = 0; double tsh = 0.0;
  while (nn >= 0) {
    int l;
    for (l = nn; l >= 1; --l) {                          // small subdiagonal element: the block l .. nn splits off
      double sdiag = fabs(A_(l - 1, l - 1)) + fabs(A_(l, l));
      if (sdiag == 0.0) sdiag = anorm;
      if (fabs(A_(l, l - 1)) + sdiag == sdiag) { A_(l, l - 1) = 0.0; break; }
    }
    double x = A_(nn, nn);
    if (l == nn) {                                       // one real eigenvalue
      rho = fmax(rho, fabs(x + tsh)); nn -= 1; its = 0; continue;
    }
    double y = A_(nn - 1, nn - 1), w = A_(nn, nn - 1) * A_(nn - 1, nn);
    if (l == nn - 1) {                                   // a 2 x 2 block: two real eigenvalues or a complex pair
      const double p = 0.5 * (y - x), qd = p * p + w; double z = sqrt(fabs(qd));
      x += tsh;
      if (qd >= 0.0) {
        z = p + (p >= 0.0 ? fabs(z) : -fabs(z));
        double r1 = x + z, r2 = r1;
        if (z != 0.0) r2 = x - w / z;
        rho = fmax(rho, fmax(fabs(r1), fabs(r2)));
      } else {
        rho = fmax(rho, sqrt((x + p) * (x + p) + z * z));
      }
      nn -= 2; its = 0; continue;
    }
    if (its == 60) return false;
    if (its == 10 || its == 20 || its == 40) {           // exceptional shift
      tsh += x;
      for (int i = 0; i <= nn; ++i) A_(i, i) -= x;
      const double sx = fabs(A_(nn, nn - 1)) + fabs(A_(nn - 1, nn - 2));
      y = x = 0.75 * sx; w = -0.4375 * sx * sx;
    }
    ++its;
    int mm; double p = 0.0, qv = 0.0, r = 0.0, z = 0.0;
    for (mm = nn - 2; mm >= l; --mm) {                   // two consecutive small subdiagonal elements
      z = A_(mm, mm);
      r = x - z; double sv = y - z;
      p = (r * sv - w) / A_(mm + 1, mm) + A_(mm, mm + 1);
      qv = A_(mm + 1, mm + 1) - z - r - sv;
      r = A_(mm + 2, mm + 1);
      sv = fabs(p) + fabs(qv) + fabs(r);
      p /= sv; qv /= sv; r /= sv;
      if (mm == l) break;
      const double u = fabs(A_(mm, mm - 1)) * (fabs(qv) + fabs(r));
      const double v = fabs(p) * (fabs(A_(mm - 1, mm - 1)) + fabs(z) + fabs(A_(mm + 1, mm + 1)));
      if (u + v == v) break;
    }
    for (int i = mm + 2; i <= nn; ++i) { A_(i, i - 2) = 0.0; if (i != mm + 2) A_(i, i - 3) = 0.0; }
    for (int k = mm; k <= nn - 1; ++k) {                 // double QR step on rows l .. nn, columns mm .. nn
      if (k != mm) {
        p = A_(k, k - 1); qv = A_(k + 1, k - 1); r = (k != nn - 1) ? A_(k + 2, k - 1) : 0.0;
        x = fabs(p) + fabs(qv) + fabs(r);
        if (x != 0.0) { p /= x; qv /= x; r /= x; }
      }
      const double nrm = sqrt(p * p + qv * qv + r * r);
      const double sg = (p >= 0.0) ? nrm : -nrm;
      if (sg != 0.0) {
        if (k == mm) { if (l != mm) A_(k, k - 1) = -A_(k, k - 1); }
        else A_(k, k - 1) = -sg * x;
        p += sg; x = p / sg; y = qv / sg; z = r / sg; qv /= p; r /= p;
        for (int j = k; j <= nn; ++j) {                  // row modification
          double pp = A_(k, j) + qv * A_(k + 1, j);
          if (k != nn - 1) { pp += r * A_(k + 2, j); A_(k + 2, j) -= pp * z; }
          A_(k + 1, j) -= pp * y; A_(k, j) -= pp * x;
        }
        const int mmin = (nn < k + 3) ? nn : k + 3;
        for (int i = l; i <= mmin; ++i) {                // column modification
          double pp = x * A_(i, k) + y * A_(i, k + 1);
          if (k != nn - 1) { pp += z * A_(i, k + 2); A_(i, k + 2) -= pp * r; }
          A_(i, k + 1) -= pp * qv; A_(i, k) -= pp;
        }
      }
    }
  }
#undef A_
  return true;
}

__global__ __launch_bounds__(64) void tz_specrad_kernel(GainParams q) {
  __shared__ double tile[TZ_GN_TILE * TZ_GN_NMAX * TZ_GN_NMAX];
  __shared__ double slab[TZ_GN_NMAX * TZ_GN_NMAX * 64];
  const int lane = threadIdx.x, s = blockIdx.x * 64 + lane;
  const bool live = s < q.S;
  double M[TZ_GN_NMAX * TZ_GN_NMAX];
  tz_gain_accumulate(q, tile, s, live, M);
  const int n = q.n;
#pragma unroll
  for (int i = 0; i < TZ_GN_NMAX; ++i)
#pragma unroll
    for (int j = 0; j < TZ_GN_NMAX; ++j) if (i * TZ_GN_NMAX + j < n * n) slab[((i * TZ_GN_NMAX + j) << 6) + lane] = M[i * TZ_GN_NMAX + j];
  double rho = 0.0;
  const bool ok = tz_spectral_radius(slab + lane, n, rho);
  if (live) { q.val[s] = rho; q.aux[s] = ok ? 0 : 2; }
}

// One workgroup per starting point, lane = generator (round 3; round 2 had lane = starting point, one wave walking all generators
// one after the other: 1 - 6 ms for the reference's 10 starts, profiles/r3j_aux_kernel_stats.csv).  A CCP step is a Jacobi sweep --
// every sign is decided against the SAME previous M -- so the generators of a start are independent: thread g forms
// d_g = <M_prev, H_g> (entries in order: the same number whatever the thread count), updates beta_g, and adds beta_g H_g into its
// partial sum of the new M; the partial sums are folded per entry (wave reductions, then the four waves in LDS, fixed order).
__global__ __launch_bounds__(256) void tz_adversary_kernel(GainParams q) {
  __shared__ double Mprev[TZ_GN_NMAX * TZ_GN_NMAX];
  __shared__ double wsum[4][TZ_GN_NMAX * TZ_GN_NMAX];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, s = blockIdx.x, n2 = q.n * q.n;
  const double* bin = q.beta_in + (size_t)s * q.ngen;
  double* bout = q.beta_out + (size_t)s * q.ngen;
  int steps = 0;
  // pass -1 forms M of the starting point (no sign update), passes 0 .. are CCP steps
  for (int it = -1; it < q.max_iter; ++it) {
    double part[TZ_GN_NMAX * TZ_GN_NMAX];
#pragma unroll
    for (int e = 0; e < TZ_GN_NMAX * TZ_GN_NMAX; ++e) part[e] = 0.0;
    int changed = 0;
    for (int g = t; g < q.ngen; g += 256) {
      const double* h = q.H + (size_t)g * n2;
      double b = (it < 0) ? bin[g] : bout[g];
      if (it >= 0) {
        double d = 0.0;
        for (int e = 0; e < n2; ++e) d += Mprev[e] * h[e];
        const double nb = (d > 0.0) ? 1.0 : ((d < 0.0) ? -1.0 : b);       // vertex of the box maximising the linearisation
        if (nb != b) { changed = 1; b = nb; }
      }
      bout[g] = b;
#pragma unroll
      for (int e = 0; e < TZ_GN_NMAX * TZ_GN_NMAX; ++e) if (e < n2) part[e] += b * h[e];
    }
#pragma unroll
    for (int e = 0; e < TZ_GN_NMAX * TZ_GN_NMAX; ++e) {
      if (e < n2) { const double v = tz_wave_reduce<RED_SUM>(part[e]); if (lane == 0) wsum[wave][e] = v; }
    }
    const int any = __syncthreads_or(changed);                 // (also the barrier between the wave sums and their readers)
    if (t < n2) Mprev[t] = q.M0[t] + ((wsum[0][t] + wsum[1][t]) + (wsum[2][t] + wsum[3][t]));
    __syncthreads();
    if (it >= 0) { ++steps; if (!any) break; }
  }
  if (t == 0) {
    double f = 0.0;
    for (int e = 0; e < n2; ++e) f += Mprev[e] * Mprev[e];
    q.val[s] = sqrt(f); q.aux[s] = steps;
  }
}
