// K1g -- literal stacked-generator tubes on the device (the general path of the reference's zonotope algebra).
//
// Reference: every Ze[k] of build_problem / build_problem_simplified (tzddpc/tzddpc.py:172-207, :283-324) is a zonotope whose
// generator columns are affine in ONE source vector each,  g = m0 + M xi_src  (xi_src = e0 or zeta_j = [xbar_j; v_j], see
// tzddpc_amd/genstack.py), and the constraints use its interval hull (:191-197):
//     centre_k,    rad^x_k = sum_g |g|,    rad^u_k = sum_g |K g| .
// The stack (up to ~1e5 generators of n (1 + n + m) doubles: tens of MB, it grows by (gamma_K + 1) per product) is streamed
// from HBM once per tile of 256 trajectories:
//   * the host sorts the generators of a tube by source and cuts them into CHUNKS (same tube, same source, <= TZ_GS_CHUNK
//     generators); one workgroup = one chunk x 256 trajectories, lane = trajectory;
//   * the chunk travels global -> LDS in tiles of TZ_GS_TILE generators with 16-byte loads per lane (fields interleaved per
//     generator, so the copy is one contiguous, fully coalesced stream);
//   * every lane keeps ITS trajectory's source vector in registers and walks the tile: the generator entries are LDS broadcast
//     reads (all lanes the same address), n (n + m) FMAs, |.| accumulated in registers -- no cross-lane reduction at all;
//   * per (chunk, trajectory) partial sums go to HBM, a second small kernel adds them per tube in chunk order (deterministic)
//     and forms the centres.
// tz_genstack_values_kernel writes the generator columns themselves (literal order) for one tube: the Ze[1] of solve() (:377).
#pragma once

#define TZ_GS_TILE 64            // generators per LDS tile
#define TZ_GS_CHUNK 1024         // generators per workgroup (host plan)

struct GsChunk { int seg, src, g0, g1; };        // generators [g0, g1) of the SORTED stack: tube seg, source src (-1 none, 0 e0, 1 + j zeta_j)

struct GenstackParams {
  int B, n, m, N, nseg, nchunk, rec;             // rec = doubles per generator record: n (1 + n + m)
  const double* recs;                            // sorted stack: G x rec, record = [m0 (n) | M row-major (n x (n+m))]
  const GsChunk* chunks;
  const double* K;                               // m x n
  const double* e0;                              // B x n
  const double* zeta;                            // B x N x (n+m)
  double* partial;                               // nchunk x B x (n+m)
};

// NCT, MCT: dim_x, dim_u known at compile time (the reference's systems: (2,1), (4,1), (5,1)); 0, 0: any n <= TZ_NMAX, m <= TZ_MMAX
// (loops run to the maxima with wave-uniform guards, strides taken from the parameters)
template <int NCT, int MCT>
__global__ __launch_bounds__(256) void tz_genstack_kernel(GenstackParams q) {
  constexpr bool GEN = (NCT == 0);
  constexpr int NC = GEN ? TZ_NMAX : NCT, MC = GEN ? TZ_MMAX : MCT, PCM = NC + MC;
  const int n = GEN ? q.n : NCT, m = GEN ? q.m : MCT, PC = n + m, REC = n * (1 + PC);
  __shared__ double tile[TZ_GS_TILE * NC * (1 + PCM)];
  __shared__ double Ks[MC * NC];
  const GsChunk ch = q.chunks[blockIdx.x];
  const int t = threadIdx.x, b = blockIdx.y * 256 + t;
  const bool live = b < q.B;
  for (int i = t; i < m * n; i += 256) Ks[i] = q.K[i];
  double xi[PCM];
#pragma unroll
  for (int c = 0; c < PCM; ++c) xi[c] = 0.0;
  if (live && ch.src == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) if (c < n) xi[c] = q.e0[(size_t)b * n + c];
  } else if (live && ch.src > 0) {
#pragma unroll
    for (int c = 0; c < PCM; ++c) if (c < PC) xi[c] = q.zeta[((size_t)b * q.N + (ch.src - 1)) * PC + c];
  }
  double ax[NC], au[MC];
#pragma unroll
  for (int i = 0; i < NC; ++i) ax[i] = 0.0;
#pragma unroll
  for (int j = 0; j < MC; ++j) au[j] = 0.0;
  for (int g0 = ch.g0; g0 < ch.g1; g0 += TZ_GS_TILE) {
    const int ng = min(TZ_GS_TILE, ch.g1 - g0);
    __syncthreads();                                             // the previous tile has been consumed
    {
      // contiguous copy of ng records: 16 bytes per lane and pass (records are 8-byte aligned doubles; the stack base is 16-byte
      // aligned and REC * TZ_GS_TILE is even, so every tile starts on a 16-byte boundary when REC is even; odd REC: 8-byte path)
      const double* srcp = q.recs + (size_t)g0 * REC;
      const int nd = ng * REC;
      if (((REC * TZ_GS_TILE) & 1) == 0 && ((((size_t)g0 * REC) & 1) == 0)) {
        const double2* s2 = reinterpret_cast<const double2*>(srcp);
        double2* d2 = reinterpret_cast<double2*>(tile);
        for (int i = t; i < (nd >> 1); i += 256) d2[i] = s2[i];
        if ((nd & 1) && t == 0) tile[nd - 1] = srcp[nd - 1];
      } else {
        for (int i = t; i < nd; i += 256) tile[i] = srcp[i];
      }
    }
    __syncthreads();
    for (int g = 0; g < ng; ++g) {
      const double* r = tile + g * REC;                          // same address in every lane: LDS broadcast
      double y[NC];
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        y[i] = 0.0;
        if (GEN && i >= n) continue;
        double a = r[i];
#pragma unroll
        for (int c = 0; c < PCM; ++c) { if (GEN && c >= PC) continue; a += r[n + i * PC + c] * xi[c]; }
        y[i] = a; ax[i] += fabs(a);
      }
#pragma unroll
      for (int j = 0; j < MC; ++j) {
        if (GEN && j >= m) continue;
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < NC; ++i) { if (GEN && i >= n) continue; a += Ks[j * n + i] * y[i]; }
        au[j] += fabs(a);
      }
    }
  }
  if (live) {
    double* o = q.partial + ((size_t)blockIdx.x * q.B + b) * PC;
#pragma unroll
    for (int i = 0; i < NC; ++i) if (i < n) o[i] = ax[i];
#pragma unroll
    for (int j = 0; j < MC; ++j) if (j < m) o[n + j] = au[j];
  }
}

struct GsReduceParams {
  int B, n, m, N, nseg;
  const int* seg_chunk_ptr;                      // nseg + 1: chunks of tube k are [ptr[k], ptr[k+1])
  const double* partial;
  const double* c0; const double* cE; const double* cZ;   // nseg x n, nseg x n x n, nseg x N x n x (n+m) (cZ may be null: all zero)
  const double* e0; const double* zeta;
  double* center; double* radx; double* radu;    // B x nseg x n, B x nseg x n, B x nseg x m
};

__global__ void tz_genstack_reduce_kernel(GsReduceParams q) {
  const int p = q.n + q.m;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)q.B * q.nseg * p) return;
  const int c = (int)(gid % p), k = (int)((gid / p) % q.nseg), b = (int)(gid / ((size_t)p * q.nseg));
  double a = 0.0;
  for (int chn = q.seg_chunk_ptr[k]; chn < q.seg_chunk_ptr[k + 1]; ++chn) a += q.partial[((size_t)chn * q.B + b) * p + c];   // fixed order
  if (c < q.n) {
    q.radx[((size_t)b * q.nseg + k) * q.n + c] = a;
    double ce = q.c0[k * q.n + c];
    for (int j = 0; j < q.n; ++j) ce += q.cE[((size_t)k * q.n + c) * q.n + j] * q.e0[(size_t)b * q.n + j];
    if (q.cZ)
      for (int j = 0; j < q.N; ++j)
        for (int cc = 0; cc < p; ++cc) ce += q.cZ[(((size_t)k * q.N + j) * q.n + c) * p + cc] * q.zeta[((size_t)b * q.N + j) * p + cc];
    q.center[((size_t)b * q.nseg + k) * q.n + c] = ce;
  } else {
    q.radu[((size_t)b * q.nseg + k) * q.m + (c - q.n)] = a;
  }
}

// Generator columns of one tube in the reference's order: Z[b][i][0] = centre_i, Z[b][i][1 + g] = (m0 + M xi_src)_i.
struct GsValuesParams {
  int B, n, m, N, seg, ngen, rec;
  const double* recs;                            // LITERAL-order stack, records of tube `seg` start at recs
  const int* src;                                // ngen
  const double* c0; const double* cE; const double* cZ;   // of this tube: n, n x n, N x n x (n+m) (may be null)
  const double* e0; const double* zeta;
  double* Z;                                     // B x n x (1 + ngen)
};

__global__ void tz_genstack_values_kernel(GsValuesParams q) {
  const int p = q.n + q.m, cols = 1 + q.ngen;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)q.B * cols) return;
  const int col = (int)(gid % cols), b = (int)(gid / cols);
  double* out = q.Z + (size_t)b * q.n * cols + col;
  if (col == 0) {
    for (int i = 0; i < q.n; ++i) {
      double ce = q.c0[i];
      for (int j = 0; j < q.n; ++j) ce += q.cE[i * q.n + j] * q.e0[(size_t)b * q.n + j];
      if (q.cZ)
        for (int j = 0; j < q.N; ++j)
          for (int cc = 0; cc < p; ++cc) ce += q.cZ[((size_t)j * q.n + i) * p + cc] * q.zeta[((size_t)b * q.N + j) * p + cc];
      out[(size_t)i * cols] = ce;
    }
    return;
  }
  const int g = col - 1, s = q.src[g];
  const double* r = q.recs + (size_t)g * q.rec;
  const double* xi = s == 0 ? q.e0 + (size_t)b * q.n : (s > 0 ? q.zeta + ((size_t)b * q.N + (s - 1)) * p : nullptr);
  const int w = s == 0 ? q.n : p;
  for (int i = 0; i < q.n; ++i) {
    double a = r[i];
    if (xi) for (int c = 0; c < w; ++c) a += r[q.n + i * p + c] * xi[c];
    out[(size_t)i * cols] = a;
  }
}

// Literal problems (dense generators): theta's tube block (c_k, rho^x_k, rho^u_k)_{k<N} taken from the evaluation of the stack of
// decision-independent generators (tz_problem_attach_tube_stack) instead of the collapsed recursion of tz_tube_kernel.
struct ThetaStackParams {
  int B, n, m, N, nseg, ntheta;
  const double* center; const double* radx; const double* radu;     // B x nseg x n, B x nseg x n, B x nseg x m
  double* theta;                                                    // B x ntheta
};

__global__ void tz_theta_stack_kernel(ThetaStackParams q) {
  const int blk = 2 * q.n + q.m;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)q.B * q.N * blk) return;
  const int c = (int)(gid % blk), k = (int)((gid / blk) % q.N), b = (int)(gid / ((size_t)blk * q.N));
  double v;
  if (c < q.n) v = q.center[((size_t)b * q.nseg + k) * q.n + c];
  else if (c < 2 * q.n) v = q.radx[((size_t)b * q.nseg + k) * q.n + (c - q.n)];
  else v = q.radu[((size_t)b * q.nseg + k) * q.m + (c - 2 * q.n)];
  q.theta[(size_t)b * q.ntheta + 2 * q.n + (size_t)k * blk + c] = v;
}
