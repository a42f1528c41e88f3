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
#ifndef TZ_GS_TILE_DOUBLES
#define TZ_GS_TILE_DOUBLES 1536  // doubles per shared LDS tile (x 2 buffers); 3072 -> 1536: 130 -> 104 registers, 48 -> 22 KB of LDS, 3 -> 4 waves per SIMD:
                                 // 1.250 -> 1.173 ms at 1024 trajectories (0.54 -> 0.57 of the f64 peak); 768: 1.184 ms
#endif
#ifndef TZ_GS_GWDIV
#define TZ_GS_GWDIV 16           // narrow kernel: a wave tile is 1 / TZ_GS_GWDIV of the shared tile; smaller tiles = fewer registers and less LDS = more
                                 // waves per SIMD to cover the LDS latency of the A reads (32 trajectories: 4 -> 0.0624 ms, 8 -> 0.0592, 16 -> 0.0587)
#endif
#define TZ_GS_MAXSUB 8           // matrix-core kernel, few trajectories: at most this many blocks share a chunk

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
  constexpr int GTILE = GEN ? 8 : TZ_GS_TILE;       // generators per LDS tile (the any-size instance: 8 x 16 x 25 doubles = 25 KB)
  const int n = GEN ? q.n : NCT, m = GEN ? q.m : MCT, PC = n + m, REC = n * (1 + PC);
  __shared__ double tile[GTILE * NC * (1 + PCM)];
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
  for (int g0 = ch.g0; g0 < ch.g1; g0 += GTILE) {
    const int ng = min(GTILE, ch.g1 - g0);
    __syncthreads();                                             // the previous tile has been consumed
    {
      // contiguous copy of ng records: 16 bytes per lane and pass (records are 8-byte aligned doubles; the stack base is 16-byte
      // aligned and REC * TZ_GS_TILE is even, so every tile starts on a 16-byte boundary when REC is even; odd REC: 8-byte path)
      const double* srcp = q.recs + (size_t)g0 * REC;
      const int nd = ng * REC;
      if (((REC * GTILE) & 1) == 0 && ((((size_t)g0 * REC) & 1) == 0)) {
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

// ---- K1g on the matrix cores ---------------------------------------------------------------------------------------------------
// Per chunk (one tube, one source) the hulls are |m0 + M Xi| summed over the generators, Xi = the source vectors of a tile of
// trajectories: a [generator rows] x [n + m] x [trajectories] product with a tiny inner dimension.  With the m rows K M, K m0
// appended to every generator on the host (P = n + m rows per generator, inner dimension P as well) rad^u comes out of the same
// product and the kernel is v_mfma_f64_4x4x4 + one v_add_f64(|.|) per matrix instruction:
//   * A operand = 4 GENERATORS x 4 inner entries of ONE component c (so the four rows of a tile are summed into the same
//     accumulator); the four blocks of the instruction take the same A (LDS broadcast read) and four groups of 4 TRAJECTORIES
//     (B operand = Xi, loaded once per workgroup, zero beyond the source's width: whatever finite number the A read picks up in
//     the padding of the inner dimension is multiplied by 0); C operand = m0 of the 4 generators;
//   * D = |.| -> acc[c][trajectory group]: P x NQ f64 accumulators per lane, folded over the 4 generator lanes (lane bits 4, 5)
//     once at the end -- fixed order, deterministic;
//   * the stack is laid out by the host exactly as the LDS tile ([group of 4 generators][component][generator][inner] then the
//     m0 block), streamed global -> registers -> LDS double-buffered: the loads of tile t + 1 are in flight while tile t is
//     multiplied, ONE barrier per tile;
//   * SPLIT = false: a workgroup covers 256 trajectories (wave w: 64 w .. 64 w + 63, NQ = 4 groups of 16) and every wave walks all
//     generators of the chunk; SPLIT = true (few trajectories: the stack is streamed once, the HBM-bound regime): all four waves
//     take the same <= 64 trajectories and every fourth group of generators each, partial sums meet in LDS in wave order;
//   * blockIdx -> (chunk, trajectory tile) so that the tiles of one chunk run on the same XCD (shared L2) back to back.
struct GsChunkM { int seg, src, q0, nq; };       // groups [q0, q0 + nq) of 4 generators each (zero-padded), tube seg, source src

struct GenstackMParams {
  int B, n, m, N, nchunk, ntt, nsub;             // ntt = blocks per chunk: trajectory tiles of 256, or (SPLIT) nsub sub-ranges of its tiles
  const double* recs;                            // groups of GD = 4 R (P + 1) doubles (R = P: K rows appended by the host; R = n: formed in the kernel)
  const double* K;                               // m x n (used when R < P)
  const GsChunkM* chunks;
  const double* e0;                              // B x n
  const double* zeta;                            // B x N x (n+m)
  double* partial;                               // nchunk x B x (n+m)
};

template <int R, int P> struct GsTile {
  static constexpr int GD = 4 * R * (P + 1);                                      // doubles per group of 4 generators
  static constexpr int GT = (TZ_GS_TILE_DOUBLES / GD >= 64) ? 64 : (TZ_GS_TILE_DOUBLES / GD >= 32) ? 32 : (TZ_GS_TILE_DOUBLES / GD >= 16) ? 16 : (TZ_GS_TILE_DOUBLES / GD >= 8) ? 8 : 4;   // groups per tile
  static constexpr int ND2 = GT * GD / 2;                                         // double2 per tile
  static constexpr int LD = (ND2 + 255) / 256;                                    // double2 per thread and tile
};

// R stored rows per generator, inner dimension P = n + m.  R = P: the m rows K M, K m0 come appended from the host.  R = n (one input:
// the reference's systems): the stack holds only [m0 | M] -- 1 / (n + 1) fewer bytes and matrix instructions -- and rad^u = sum |K g| is formed
// from the n results of a generator group, which sit in the same lanes (one fused multiply-add per component and trajectory group).
template <int R, int P, int NQ, bool SPLIT>
__global__ __launch_bounds__(256) void tz_genstack_mfma_kernel(GenstackMParams q) {
  constexpr int KS = (P + 3) / 4, GD = GsTile<R, P>::GD, GT = GsTile<R, P>::GT, LD = GsTile<R, P>::LD, MK = P - R;
  double kr[MK > 0 ? MK : 1][R];                 // K (uniform: scalar registers)
#pragma unroll
  for (int jj = 0; jj < MK; ++jj)
#pragma unroll
    for (int c = 0; c < R; ++c) kr[jj][c] = q.K[jj * R + c];
  __shared__ double2 tile2[2][GT * GD / 2 + 2];                   // +2: the last A read of a tile may run 3 doubles past the group
  __shared__ double xred[SPLIT ? 3 * P * NQ * 16 : 1];
  // (chunk, trajectory tile) of this block: the ntt tiles of a chunk are 8 blocks apart -> same XCD, consecutive waves of blocks
  const int bid = blockIdx.x;
  const int cgrp = bid / (8 * q.ntt), rem = bid % (8 * q.ntt);
  const int chunk = cgrp * 8 + (rem & 7);
  const int tt = SPLIT ? 0 : rem >> 3, sub = SPLIT ? rem >> 3 : 0;      // SPLIT: q.ntt counts the sub-blocks of a chunk (q.nsub)
  if (chunk >= q.nchunk) return;
  const GsChunkM ch = q.chunks[chunk];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int tb = SPLIT ? tt * 16 * NQ : tt * 256 + wave * 64;     // first trajectory of this wave
  // B operands: lane (k = lane >> 4, trajectory lane & 15 of group qq) holds Xi[4 s + k] of its trajectory
  double bx[KS][NQ];
  {
    const int k = lane >> 4, n = q.n, pc = q.n + q.m;
    const int width = ch.src == 0 ? n : (ch.src > 0 ? pc : 0);
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
      const int b = tb + 16 * qq + (lane & 15);
      const double* sv = ch.src == 0 ? q.e0 + (size_t)b * n : q.zeta + ((size_t)b * q.N + (ch.src > 0 ? ch.src - 1 : 0)) * pc;
#pragma unroll
      for (int s = 0; s < KS; ++s) bx[s][qq] = (b < q.B && 4 * s + k < width) ? sv[4 * s + k] : 0.0;
    }
  }
  double acc[P][NQ];
#pragma unroll
  for (int c = 0; c < P; ++c)
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) acc[c][qq] = 0.0;
  const double2* src2 = reinterpret_cast<const double2*>(q.recs + (size_t)ch.q0 * GD);
  // sub-range of the chunk's tiles this block walks (SPLIT only: nsub blocks share a chunk so that there are enough blocks to
  // balance the chip; their partial sums are separate rows of `partial`, added by the reduce kernel in order)
  const int ntile_all = (ch.nq + GT - 1) / GT;
  const int t_lo = SPLIT ? (int)(((long long)ntile_all * sub) / q.nsub) : 0;
  const int t_hi = SPLIT ? (int)(((long long)ntile_all * (sub + 1)) / q.nsub) : ntile_all;
  const int ntile = t_hi - t_lo;
  double2 stA[LD], stB[LD];                                      // two tiles in flight: HBM latency is longer than one tile's products
  auto fetch = [&](double2 (&stage)[LD], int tl) {               // global -> registers (tile t_lo + tl), nothing waits for it here
    const int tg = t_lo + tl;
    const int nd2 = min(GT, ch.nq - tg * GT) * (GD / 2);
    const double2* s = src2 + (size_t)tg * (GT * GD / 2);
#pragma unroll
    for (int i = 0; i < LD; ++i) { const int e = t + 256 * i; stage[i] = (e < nd2) ? s[e] : make_double2(0.0, 0.0); }
  };
  auto park = [&](const double2 (&stage)[LD], int buf) {         // registers -> LDS
#pragma unroll
    for (int i = 0; i < LD; ++i) { const int e = t + 256 * i; if (e < GT * GD / 2) tile2[buf][e] = stage[i]; }
  };
  const int ai = (lane & 3) * P + (lane >> 4);                   // A: generator i = lane & 3, inner entry k = lane >> 4 (+ 4 s)
  const int ci = 4 * R * P + (lane >> 4);                        // C (D layout): generator i = lane >> 4
  auto products = [&](int tl) {
    const double* buf = reinterpret_cast<const double*>(tile2[tl & 1]);
    const int ng = min(GT, ch.nq - (t_lo + tl) * GT);
    for (int g = SPLIT ? wave : 0; g < ng; g += SPLIT ? 4 : 1) {
      const double* gb = buf + g * GD;
      double ku[MK > 0 ? MK : 1][NQ];
#pragma unroll
      for (int jj = 0; jj < MK; ++jj)
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) ku[jj][qq] = 0.0;
#pragma unroll
      for (int c = 0; c < R; ++c) {
        const double m0v = gb[ci + 4 * c];
        double d[NQ];
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) d[qq] = m0v;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const double a = gb[c * 4 * P + ai + 4 * s];
#pragma unroll
          for (int qq = 0; qq < NQ; ++qq) d[qq] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bx[s][qq], d[qq], 0, 0, 0);
        }
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
          acc[c][qq] += fabs(d[qq]);
#pragma unroll
          for (int jj = 0; jj < MK; ++jj) ku[jj][qq] = __builtin_fma(kr[jj][c], d[qq], ku[jj][qq]);
        }
      }
#pragma unroll
      for (int jj = 0; jj < MK; ++jj)
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) acc[R + jj][qq] += fabs(ku[jj][qq]);
    }
  };
  if (t < 2) { tile2[0][GT * GD / 2 + t] = make_double2(0.0, 0.0); tile2[1][GT * GD / 2 + t] = make_double2(0.0, 0.0); }
  if (ntile > 0) {
    fetch(stA, 0);
    if (ntile > 1) fetch(stB, 1);
    park(stA, 0);
    __syncthreads();
    for (int tl = 0; tl < ntile; tl += 2) {                      // two tiles per trip: the register sets alternate statically
      if (tl + 2 < ntile) fetch(stA, tl + 2);
      products(tl);
      if (tl + 1 < ntile) park(stB, 1);
      __syncthreads();
      if (tl + 1 < ntile) {
        if (tl + 3 < ntile) fetch(stB, tl + 3);
        products(tl + 1);
        if (tl + 2 < ntile) park(stA, 0);
        __syncthreads();
      }
    }
  }
  // fold the four generator lanes (lane bits 4 and 5), fixed order
#pragma unroll
  for (int c = 0; c < P; ++c)
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
      double v = acc[c][qq];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      acc[c][qq] = v;
    }
  if (SPLIT) {
    if (wave > 0 && lane < 16) {
#pragma unroll
      for (int c = 0; c < P; ++c)
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) xred[(((wave - 1) * P + c) * NQ + qq) * 16 + lane] = acc[c][qq];
    }
    __syncthreads();
    if (wave == 0 && lane < 16) {
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) {
        const int b = tb + 16 * qq + lane;
#pragma unroll
        for (int c = 0; c < P; ++c) {
          double v = acc[c][qq];
          for (int w = 0; w < 3; ++w) v += xred[((w * P + c) * NQ + qq) * 16 + lane];
          if (b < q.B) q.partial[((size_t)(chunk * q.nsub + sub) * q.B + b) * P + c] = v;
        }
      }
    }
  } else if (lane < 16) {
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
      const int b = tb + 16 * qq + lane;
      if (b < q.B) {
#pragma unroll
        for (int c = 0; c < P; ++c) q.partial[((size_t)(chunk * q.nsub + sub) * q.B + b) * P + c] = acc[c][qq];
      }
    }
  }
}

// Few trajectories (<= 64), wave-private pipeline: every wave streams ITS tiles (GW = GT / 4 groups each, dealt round-robin to the
// four waves) through its own double buffer in LDS -- global -> registers -> LDS -> A operands -- with wave-level ordering only: no
// workgroup barrier inside the stream, so one wave's loads, another's LDS writes and a third's matrix instructions overlap instead
// of meeting at a barrier per tile (the SPLIT form of tz_genstack_mfma_kernel: MFMA pipe 44 % busy, HBM at half its achievable rate,
// neither saturated).  Same arithmetic per wave as there; the four waves' sums meet in LDS in wave order at the end.
template <int R, int P, int NQ>
__global__ __launch_bounds__(256) void tz_genstack_mfma_narrow_kernel(GenstackMParams q) {
  constexpr int KS = (P + 3) / 4, GD = GsTile<R, P>::GD, GW = (GsTile<R, P>::GT >= TZ_GS_GWDIV) ? GsTile<R, P>::GT / TZ_GS_GWDIV : 1, MK = P - R;
  double kr[MK > 0 ? MK : 1][R];
#pragma unroll
  for (int jj = 0; jj < MK; ++jj)
#pragma unroll
    for (int c = 0; c < R; ++c) kr[jj][c] = q.K[jj * R + c];
  constexpr int WD2 = GW * GD / 2, LDW = (WD2 + 63) / 64;
  __shared__ double2 wtile[4][2][WD2 + 2];
  __shared__ double xred[3 * P * NQ * 16];
  const int bid = blockIdx.x;
  const int cgrp = bid / (8 * q.ntt), rem = bid % (8 * q.ntt);
  const int chunk = cgrp * 8 + (rem & 7), sub = rem >> 3;
  if (chunk >= q.nchunk) return;
  const GsChunkM ch = q.chunks[chunk];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double bx[KS][NQ];
  {
    const int k = lane >> 4, n = q.n, pc = q.n + q.m;
    const int width = ch.src == 0 ? n : (ch.src > 0 ? pc : 0);
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
      const int b = 16 * qq + (lane & 15);
      const double* sv = ch.src == 0 ? q.e0 + (size_t)b * n : q.zeta + ((size_t)b * q.N + (ch.src > 0 ? ch.src - 1 : 0)) * pc;
#pragma unroll
      for (int s = 0; s < KS; ++s) bx[s][qq] = (b < q.B && 4 * s + k < width) ? sv[4 * s + k] : 0.0;
    }
  }
  double acc[P][NQ];
#pragma unroll
  for (int c = 0; c < P; ++c)
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) acc[c][qq] = 0.0;
  const double2* src2 = reinterpret_cast<const double2*>(q.recs + (size_t)ch.q0 * GD);
  // wave tiles of this block: [t_lo, t_hi) of the chunk's ceil(nq / GW); this wave takes t_lo + wave, + 4, ...
  const int nwt_all = (ch.nq + GW - 1) / GW;
  const int t_lo = (int)(((long long)nwt_all * sub) / q.nsub), t_hi = (int)(((long long)nwt_all * (sub + 1)) / q.nsub);
  const int mine = (t_hi - t_lo - wave + 3) / 4;                 // number of tiles of this wave (<= 0: none)
  double2 stA[LDW], stB[LDW];
  auto fetch = [&](double2 (&stage)[LDW], int j) {
    const int tg = t_lo + wave + 4 * j;
    const int nd2 = min(GW, ch.nq - tg * GW) * (GD / 2);
    const double2* s = src2 + (size_t)tg * WD2;
#pragma unroll
    for (int i = 0; i < LDW; ++i) { const int e = lane + 64 * i; stage[i] = (e < nd2) ? s[e] : make_double2(0.0, 0.0); }
  };
  auto park = [&](const double2 (&stage)[LDW], int buf) {
#pragma unroll
    for (int i = 0; i < LDW; ++i) { const int e = lane + 64 * i; if (e < WD2) wtile[wave][buf][e] = stage[i]; }
  };
  auto wsync = [&]() {                                           // this wave's LDS writes before its later reads (and the reverse)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const int ai = (lane & 3) * P + (lane >> 4), ci = 4 * R * P + (lane >> 4);
  auto products = [&](int j) {
    const double* buf = reinterpret_cast<const double*>(wtile[wave][j & 1]);
    const int ng = min(GW, ch.nq - (t_lo + wave + 4 * j) * GW);
    for (int g = 0; g < ng; ++g) {
      const double* gb = buf + g * GD;
      double ku[MK > 0 ? MK : 1][NQ];
#pragma unroll
      for (int jj = 0; jj < MK; ++jj)
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) ku[jj][qq] = 0.0;
#pragma unroll
      for (int c = 0; c < R; ++c) {
        const double m0v = gb[ci + 4 * c];
        double d[NQ];
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) d[qq] = m0v;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const double a = gb[c * 4 * P + ai + 4 * s];
#pragma unroll
          for (int qq = 0; qq < NQ; ++qq) d[qq] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bx[s][qq], d[qq], 0, 0, 0);
        }
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
          acc[c][qq] += fabs(d[qq]);
#pragma unroll
          for (int jj = 0; jj < MK; ++jj) ku[jj][qq] = __builtin_fma(kr[jj][c], d[qq], ku[jj][qq]);
        }
      }
#pragma unroll
      for (int jj = 0; jj < MK; ++jj)
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) acc[R + jj][qq] += fabs(ku[jj][qq]);
    }
  };
  if (lane < 2) { wtile[wave][0][WD2 + lane] = make_double2(0.0, 0.0); wtile[wave][1][WD2 + lane] = make_double2(0.0, 0.0); }
  if (mine > 0) {
    fetch(stA, 0);
    if (mine > 1) fetch(stB, 1);
    park(stA, 0);
    wsync();
    for (int j = 0; j < mine; j += 2) {
      if (j + 2 < mine) fetch(stA, j + 2);
      products(j);
      if (j + 1 < mine) park(stB, 1);
      wsync();
      if (j + 1 < mine) {
        if (j + 3 < mine) fetch(stB, j + 3);
        products(j + 1);
        if (j + 2 < mine) park(stA, 0);
        wsync();
      }
    }
  }
#pragma unroll
  for (int c = 0; c < P; ++c)
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
      double v = acc[c][qq];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      acc[c][qq] = v;
    }
  if (wave > 0 && lane < 16) {
#pragma unroll
    for (int c = 0; c < P; ++c)
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) xred[(((wave - 1) * P + c) * NQ + qq) * 16 + lane] = acc[c][qq];
  }
  __syncthreads();
  if (wave == 0 && lane < 16) {
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
      const int b = 16 * qq + lane;
#pragma unroll
      for (int c = 0; c < P; ++c) {
        double v = acc[c][qq];
        for (int w = 0; w < 3; ++w) v += xred[((w * P + c) * NQ + qq) * 16 + lane];
        if (b < q.B) q.partial[((size_t)(chunk * q.nsub + sub) * q.B + b) * P + c] = v;
      }
    }
  }
}

struct GsReduceParams {
  int B, n, m, N, nseg, nsub;                    // nsub partial rows per chunk (1 unless the matrix-core kernel split the chunks)
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
  for (int chn = q.seg_chunk_ptr[k] * q.nsub; chn < q.seg_chunk_ptr[k + 1] * q.nsub; ++chn) a += q.partial[((size_t)chn * q.B + b) * p + c];   // fixed order
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

// The same for few trajectories: sixteen lanes per output.  One thread per output walks its tube's ~60 partial rows and the N (n + m)
// centre terms one dependent round trip after the other -- 26 us at 32 trajectories, half of what the stream kernel itself takes.  Here
// lane l of a group of 16 takes the rows l, l + 16, ... (and the centre terms likewise) and the sixteen sums are folded by a fixed
// xor tree: deterministic (the same tree every run), four loads deep.
__global__ __launch_bounds__(256) void tz_genstack_reduce16_kernel(GsReduceParams q) {
  const int p = q.n + q.m;
  const size_t gid = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  const bool live = gid < (size_t)q.B * q.nseg * p;
  const size_t g2 = live ? gid : 0;
  const int c = (int)(g2 % p), k = (int)((g2 / p) % q.nseg), b = (int)(g2 / ((size_t)p * q.nseg));
  double a = 0.0, ce = 0.0;
  const int ch0 = q.seg_chunk_ptr[k] * q.nsub, ch1 = q.seg_chunk_ptr[k + 1] * q.nsub;
  for (int chn = ch0 + l; chn < ch1; chn += 16) a += q.partial[((size_t)chn * q.B + b) * p + c];
  if (c < q.n) {
    if (l == 0) ce = q.c0[k * q.n + c];
    if (l < q.n) ce += q.cE[((size_t)k * q.n + c) * q.n + l] * q.e0[(size_t)b * q.n + l];
    if (q.cZ) {
      const int nt = q.N * p;
      const double* cz = q.cZ + ((size_t)k * q.N * q.n) * p;      // [j][c][cc]
      const double* zt = q.zeta + (size_t)b * q.N * p;              // [j][cc]
      for (int e = l; e < nt; e += 16) { const int j = e / p, cc = e - j * p; ce += cz[((size_t)j * q.n + c) * p + cc] * zt[e]; }
    }
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 16); ce += __shfl_xor(ce, off, 16); }
  if (live && l == 0) {
    if (c < q.n) { q.radx[((size_t)b * q.nseg + k) * q.n + c] = a; q.center[((size_t)b * q.nseg + k) * q.n + c] = ce; }
    else q.radu[((size_t)b * q.nseg + k) * q.m + (c - q.n)] = a;
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
