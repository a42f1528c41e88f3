// Newton-system routines of tz_ipm_kernel for problems with more than 64 variables (17 .. 64 tile columns): Gram matrix,
// Cholesky factorisation and triangular solves on the "tile triangle" layout of H.
//
//   H(4I + i, 4J + j), J <= I, lives at  Ht[(I (I + 1) / 2 + J) * TS + 4 i + j]     (TS = 17 doubles per tile when the
//   problem leaves room in LDS, else 16: the odd stride spreads the tile rows that one LDS access touches over the banks)
//
// Exactly the lower-triangular tiles are stored -- the quad layout of the small problems (tz_ipm.hip.h) wastes the upper tiles
// of every diagonal quad and pads, which is what kept the double integrator at horizon 80 (158 variables) out of LDS.
// Included by tz_ipm.hip.h (uses its DPP / pinned-load / pivot helpers).
#pragma once

__device__ inline int tz_tri(int I) { return (I * (I + 1)) >> 1; }

// ---------------------------------------------------------------------------------------------------------------------------
// Gram matrix  H = P + G' diag(w) G + reg I  by v_mfma_f64_4x4x4, register-blocked.
//
// Unit of work: the tiles (I, J) of  tile rows [ib, ib + nr) x tile columns [jb, jb + nc),  nr, nc <= TZ_GU, J <= I; a wave keeps
// the whole unit in accumulators and streams G once over all super-steps (16 rows of G).  As in the small-problem Gram the
// four blocks of the instruction are four patch rows of the SAME output tile: lane (k, blk, ij) loads element [k][ij] of tile T of
// patch row 4 s + blk -- one value per (super-step, tile) that serves as the A operand (times w) of tile row T and as the B
// operand of tile column T -- so a unit costs nr + nc loads per nr * nc MFMAs.  The library orders variables and rows so that the
// non-zeros of G lie under a staircase (tz_problem_create): a unit is dense on the super-steps [s0, S) and skipped before, no
// per-tile tests in the loop.  The host deals the units to the four waves by their MFMA count.
// ---------------------------------------------------------------------------------------------------------------------------
#define TZ_GU 8                  // largest unit any variant uses
struct TzGUnit { int ib, jb, s0; };     // tile rows [ib, ib + U) x tile columns [jb, jb + U); s0: first super-step that touches tile column ib

template <bool DIAG, int U>
struct TzGuStage { double vr[U]; double vc[DIAG ? 1 : U]; double w; };

// U: tiles per side of a unit (the host cuts the tile index range, padded with all-zero tile columns if need be, into ranges of
// exactly U tiles); NST: super-steps in flight -- the loads of super-step s + NST - 1 are issued before the products of
// super-step s (the patches come from L2, ~700 cycles away).  No masks and no per-tile tests: the staircase ordering
// (tzddpc_hip.hip) makes the unit dense on [s0, S).  P (and reg on the diagonal) enter through the initial value of the
// accumulators of block 0: their loads are in flight together with the first patches instead of in front of the stores.
template <bool DIAG, int U, int NST>
__device__ inline void tz_gram_unit(const IpmParams& p, double* Ht, const double* wv, const TzGUnit u, unsigned long long* pacc) {
  unsigned long long tq0 = pacc ? __builtin_amdgcn_s_memtime() : 0;
  const int lane = tz_tid() & 63;
  const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
  const int Tz = p.Tz, Kc = p.Kc, S = (Kc + 3) >> 2, TS = p.TS;
  const unsigned rowbytes = (unsigned)(Tz + 1) * 128u;
  const int ib = u.ib, jb = u.jb;
  double acc[U][U];
  {
    const int i = lane >> 4, j = lane & 3;                               // D layout: lane (i, blk, j)
#pragma unroll
    for (int a = 0; a < U; ++a)
#pragma unroll
      for (int b = 0; b < U; ++b) {
        if (DIAG && b > a) { acc[a][b] = 0.0; continue; }
        const int r = 4 * (ib + a) + i, c = 4 * (jb + b) + j;
        double v = 0.0;
        if (blk == 0 && r < p.nP && c < p.nP) v = p.P[(size_t)r * p.nzp + c];   // P is zero beyond its leading nP x nP block
        if (blk == 0 && r == c && r < p.nz) v += p.reg;
        acc[a][b] = v;
      }
  }
  // byte offsets of the unit's tiles inside a patch row; tile columns >= Tz (padding of the last range) read the all-zero tile Tz
  unsigned offr[U], offc[U];
#pragma unroll
  for (int a = 0; a < U; ++a) { offr[a] = (unsigned)min(ib + a, Tz) * 128u; offc[a] = (unsigned)min(jb + a, Tz) * 128u; }
  // address = wave-uniform tile base (scalar registers) + one 32-bit lane offset per super-step: no 64-bit vector address arithmetic per load
  const char* gp = (const char*)p.Gp;
  const unsigned laneoff = (unsigned)(4 * k + ij) * 8u;
  auto load = [&](int s, TzGuStage<DIAG, U>& st) {
    int kc = 4 * s + blk; kc = (kc < Kc) ? kc : Kc;                       // patch row Kc is all zero, wv[4 Kc + k] = 0
    st.w = wv[4 * kc + k];
    const unsigned voff = (unsigned)kc * rowbytes + laneoff;
#pragma unroll
    for (int a = 0; a < U; ++a) st.vr[a] = tz_ld_pinned((const double*)(gp + offr[a] + (size_t)voff));
    if (!DIAG) {
#pragma unroll
      for (int b = 0; b < U; ++b) st.vc[DIAG ? 0 : b] = tz_ld_pinned((const double*)(gp + offc[b] + (size_t)voff));
    }
  };
  auto mma = [&](const TzGuStage<DIAG, U>& st) {
#pragma unroll
    for (int a = 0; a < U; ++a) {
      const double A = st.vr[a] * st.w;
#pragma unroll
      for (int b = 0; b < U; ++b) {
        if (DIAG && b > a) continue;
        acc[a][b] = __builtin_amdgcn_mfma_f64_4x4x4f64(A, DIAG ? st.vr[b] : st.vc[DIAG ? 0 : b], acc[a][b], 0, 0, 0);
      }
    }
  };
  TzGuStage<DIAG, U> st[NST];
  const int s0 = u.s0;
#pragma unroll
  for (int d = 0; d < NST - 1; ++d) load(s0 + d, st[d]);
  for (int s = s0; s < S; s += NST) {
#pragma unroll
    for (int d = 0; d < NST; ++d) {                                       // stage d holds super-step s + d; its slot is refilled NST - 1 steps ahead
      load(s + d + NST - 1, st[(d + NST - 1) % NST]);
      mma(st[d]);
    }
  }
  if (pacc) { unsigned long long t1 = __builtin_amdgcn_s_memtime(); pacc[PH_GRAM_LOOP] += t1 - tq0; tq0 = t1; }
  // fold the four patch rows (blk) of every tile with two row rotations: every lane of a row of 16 then holds the tile sum;
  // lane (i, blk, j) then stores the tile of column jb + 4 q + blk -- four tiles per store instruction
  const int i = lane >> 4, j = lane & 3;
#pragma unroll
  for (int a = 0; a < U; ++a) {
    const int I = ib + a, r = 4 * I + i;
    if (I >= Tz) continue;
#pragma unroll
    for (int b = 0; b < U; ++b) {
      if (DIAG && b > a) continue;
      double v = acc[a][b]; v += tz_row_ror<4>(v); v += tz_row_ror<8>(v); acc[a][b] = v;
    }
#pragma unroll
    for (int q = 0; q < (U + 3) / 4; ++q) {
      if (DIAG && 4 * q > a) continue;
      const double v = tz_sel4(blk, acc[a][4 * q], (4 * q + 1 < U) ? acc[a][(4 * q + 1 < U) ? 4 * q + 1 : 0] : 0.0,
                               (4 * q + 2 < U) ? acc[a][(4 * q + 2 < U) ? 4 * q + 2 : 0] : 0.0, (4 * q + 3 < U) ? acc[a][(4 * q + 3 < U) ? 4 * q + 3 : 0] : 0.0);
      const int J = jb + 4 * q + blk;
      if (4 * q + blk < U && J <= I) {
        const int c = 4 * J + j;
        Ht[(tz_tri(I) + J) * TS + 4 * i + j] = (r == c && r >= p.nz) ? 1.0 : v;      // padding rows: unit diagonal
      }
    }
  }
  if (pacc) pacc[PH_GRAM_RED] += __builtin_amdgcn_s_memtime() - tq0;
}

// UMAX: largest unit the variant's register budget holds; the host picks U in [UMAX - 2, UMAX] per problem (p.gu)
template <int UMAX, int NST>
__device__ inline void tz_gram_tt(const IpmParams& p, double* Ht, const double* wv, unsigned long long* pacc = nullptr) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u0 = __builtin_amdgcn_readfirstlane(p.gunit_ptr[wave]), u1 = __builtin_amdgcn_readfirstlane(p.gunit_ptr[wave + 1]);
  const int gu = __builtin_amdgcn_readfirstlane(p.gu);
  for (int ui = u0; ui < u1; ++ui) {
    TzGUnit u = p.gunits[ui];
    u.ib = __builtin_amdgcn_readfirstlane(u.ib); u.jb = __builtin_amdgcn_readfirstlane(u.jb); u.s0 = __builtin_amdgcn_readfirstlane(u.s0);
    const bool dg = u.ib == u.jb;
    if (gu == UMAX) { if (dg) tz_gram_unit<true, UMAX, NST>(p, Ht, wv, u, pacc); else tz_gram_unit<false, UMAX, NST>(p, Ht, wv, u, pacc); }
    else if (gu == UMAX - 1) { if (dg) tz_gram_unit<true, UMAX - 1, NST>(p, Ht, wv, u, pacc); else tz_gram_unit<false, UMAX - 1, NST>(p, Ht, wv, u, pacc); }
    else { if (dg) tz_gram_unit<true, UMAX - 2, NST>(p, Ht, wv, u, pacc); else tz_gram_unit<false, UMAX - 2, NST>(p, Ht, wv, u, pacc); }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Cholesky  H = L L'  in place, tile-4 left-looking, two workgroup barriers per tile column:
//
//   phase A(pp)   wave 0     brings the diagonal tile (pp, pp) up to date with column pp - 1, factors it (four dependent
//                            rsqrt + Newton pivots: the one serial chain of the factorisation) and publishes the factor and its
//                            inverse (dfac, dinv[pp]);
//                 waves 1-3  meanwhile (i) finish column pp: tiles (I, pp), I > pp, minus the contribution of column pp - 1,
//                            (ii) bring column pp + 1 up to date with the columns < pp (everything but the column being
//                            factored) -- the bulk of the MFMA work, off the critical path.
//   phase B(pp)   all        panel: rows below the diagonal tile times the inverse of its factor.
//
// Operands come straight from LDS in the layout the instruction wants (A: L(4 Ia + i, 4 k2 + k'), blk = four tile rows Ia;
// B: -L(4 c + j, 4 k2 + k'), the same tile for the four blocks); consecutive k2 are TS doubles apart, the next operands are
// fetched before the current product is issued.
// ---------------------------------------------------------------------------------------------------------------------------
// tiles (I, c) for I = Ifirst + 4 g + blk of the row groups g = gfirst, gfirst + gstep, ...: minus sum_{k2 in [k0, k1)} L(I, k2) L(c, k2)'.
// Two row groups and two consecutive k2 per trip (four independent accumulator chains); the six operands of the next trip are
// requested before the products of this one are issued -- unconditionally (the offsets of the last trip are clamped, an odd
// column count ends with a zero B operand), so that the loop body is straight-line code and the wait in front of the
// products only covers the loads of the previous trip.
__device__ inline void tz_tt_update(double* Ht, int TS, int Tz, int c, int Ifirst, int k0, int k1, int gfirst, int gstep) {
  if (k1 <= k0) return;
  const int lane = tz_tid() & 63;
  const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
  const int nk = k1 - k0, ntrip = (nk + 1) >> 1, klast = nk - 1;
  const double* pb = Ht + (tz_tri(c) + k0) * TS + 4 * ij + k;                // L(4c + j, 4k2 + k'), j = ij, k' = k
  for (int g = gfirst; Ifirst + 4 * g < Tz; g += 2 * gstep) {
    const int Ia = Ifirst + 4 * g + blk, Ib = Ia + 4 * gstep;
    const bool va = Ia < Tz, vb = Ib < Tz;
    const double* pa = Ht + (tz_tri(va ? Ia : c) + k0) * TS + 4 * ij + k;    // L(4Ia + i, 4k2 + k'), i = ij
    const double* pa2 = Ht + (tz_tri(vb ? Ib : c) + k0) * TS + 4 * ij + k;
    double* pc = Ht + (tz_tri(va ? Ia : c) + c) * TS + 4 * k + ij;           // H(4Ia + i', 4c + j'), i' = k, j' = ij
    double* pc2 = Ht + (tz_tri(vb ? Ib : c) + c) * TS + 4 * k + ij;
    double a0 = va ? *pc : 0.0, a1 = 0.0, c0 = vb ? *pc2 : 0.0, c1 = 0.0;
    int o0 = 0, o1 = min(1, klast) * TS;
    if (nk == 1) {                                                          // a single column: one product per row group
      const double nb = -pb[0];
      a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa[0], nb, a0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa2[0], nb, c0, 0, 0, 0);
    } else {
      // operands two trips ahead (the LDS round trip is longer than the four products of a trip).  tz_ld_pinned: a load the
      // optimiser may not sink to its use (it would undo the prefetch: LDS reads have no side effects)
      auto ld6 = [&](int tr, double (&v)[6]) {
        const int q0 = min(2 * tr, klast) * TS, q1 = min(2 * tr + 1, klast) * TS;
        v[0] = tz_ld_pinned(pa + q0); v[1] = tz_ld_pinned(pa2 + q0); v[2] = tz_ld_pinned(pb + q0);
        v[3] = tz_ld_pinned(pa + q1); v[4] = tz_ld_pinned(pa2 + q1); v[5] = tz_ld_pinned(pb + q1);
      };
      auto mm4 = [&](int tr, const double (&v)[6]) {
        const double nb0 = -v[2], nb1 = (2 * tr + 1 < nk) ? -v[5] : 0.0;
        a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(v[0], nb0, a0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(v[1], nb0, c0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v[3], nb1, a1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v[4], nb1, c1, 0, 0, 0);
      };
      double r0[6], r1[6], r2[6];
      ld6(0, r0); ld6(1, r1);
      int tr = 0;
      for (; tr + 2 < ntrip; tr += 3) {
        ld6(tr + 2, r2); mm4(tr, r0);
        ld6(tr + 3, r0); mm4(tr + 1, r1);
        ld6(tr + 4, r1); mm4(tr + 2, r2);
      }
      if (tr < ntrip) mm4(tr, r0);
      if (tr + 1 < ntrip) mm4(tr + 1, r1);
    }
    if (va) *pc = a0 + a1;
    if (vb) *pc2 = c0 + c1;
  }
}

// Phase A of column pp for the waves that do not factor (tz_cholesky_tt), both updates in one pass over the rows below pp:
//   (i)  tiles (I, pp)     -= L(I, pp-1) L(pp, pp-1)'                    -- finishes column pp
//   (ii) tiles (I, pp + 1) -= sum_{k2 < pp} L(I, k2) L(pp+1, k2)'        -- column pp + 1 (incl. its diagonal tile) up to date with the columns < pp
// for I = pp + 1 + 4 g + blk, row groups g = gfirst, gfirst + gstep, ...  The A operands L(I, k2) are shared by the two columns;
// the C tiles and the first operands of both are requested together (one LDS round trip up front instead of two calls' worth),
// all addresses advance by constants (no integer multiplies in the loop).
__device__ inline void tz_tt_phase_a(double* Ht, int TS, int Tz, int pp, int gfirst, int gstep) {
  const int lane = tz_tid() & 63;
  const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
  const int lo = 4 * ij + k;                                               // element [ij][k] of an operand tile
  const bool two = pp + 1 < Tz;                                            // column pp + 1 exists
  const double* pb1 = Ht + (tz_tri(pp) + pp - 1) * TS + lo;                // L(4pp + j, 4(pp-1) + k')
  const double* pb2 = Ht + tz_tri(two ? pp + 1 : pp) * TS + lo;            // L(4(pp+1) + j, 4 k2 + k'), k2 = 0 ...
  for (int g = gfirst; pp + 1 + 4 * g < Tz; g += gstep) {                  // (two groups at a time was slower: most columns leave one group per wave)
    const int I = pp + 1 + 4 * g + blk;
    const bool v = I < Tz;
    const int tI = tz_tri(v ? I : pp) * TS;
    const double* pa = Ht + tI + lo;                                       // L(4I + i, 4 k2 + k'), k2 = 0 ...
    double* pc1 = Ht + tI + pp * TS + 4 * k + ij;                          // H(4I + i', 4pp + j')
    double* pc2 = pc1 + TS;                                                // H(4I + i', 4(pp+1) + j')   (I >= pp + 1: inside the triangle)
    double c1 = tz_ld_pinned(pc1), c2 = two ? tz_ld_pinned(pc2) : 0.0, c3 = 0.0;
    const double alast = tz_ld_pinned(pa + (pp - 1) * TS), b1 = tz_ld_pinned(pb1);
    if (two) {
      const int nk = pp, klast = nk - 1;
      auto ld = [&](int k2, double& a, double& b) { const int o = min(k2, klast) * TS; a = tz_ld_pinned(pa + o); b = tz_ld_pinned(pb2 + o); };
      double a0, b0, a1, b1n, a2, b2, a3, b3;
      ld(0, a0, b0); ld(1, a1, b1n); ld(2, a2, b2); ld(3, a3, b3);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(alast, -b1, c1, 0, 0, 0);
      int k2 = 0;
      for (; k2 + 3 < nk; k2 += 4) {                                       // operands four columns ahead
        const double x0 = a0, y0 = -b0, x1 = a1, y1 = -b1n, x2 = a2, y2 = -b2, x3 = a3, y3 = -b3;
        ld(k2 + 4, a0, b0); ld(k2 + 5, a1, b1n); ld(k2 + 6, a2, b2); ld(k2 + 7, a3, b3);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, y0, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, y1, c3, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, y2, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x3, y3, c3, 0, 0, 0);
      }
      if (k2 < nk) c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, -b0, c2, 0, 0, 0);
      if (k2 + 1 < nk) c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, -b1n, c3, 0, 0, 0);
      if (k2 + 2 < nk) c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, -b2, c2, 0, 0, 0);
      if (v) { *pc1 = c1; *pc2 = c2 + c3; }
    } else {
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(alast, -b1, c1, 0, 0, 0);
      if (v) *pc1 = c1;
    }
  }
}

// dfac (LDS, 10 doubles): l10 l20 l21 l30 l31 l32 i00 i11 i22 i33 of the diagonal tile being eliminated
__device__ inline bool tz_cholesky_tt(const IpmParams& p, double* Ht, double* dinv, double* dfac, int* flag, unsigned long long* pacc = nullptr) {
  const int Tz = p.Tz, TS = p.TS;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int pp = 0; pp < Tz; ++pp) {
    unsigned long long tc0 = pacc ? __builtin_amdgcn_s_memtime() : 0;
    if (wave == 0) {
      __builtin_amdgcn_s_setprio(TZ_PRIO);
      const double* d = Ht + (tz_tri(pp) + pp) * TS;
      double a00 = d[0], a10 = d[4], a11 = d[5], a20 = d[8], a21 = d[9], a22 = d[10], a30 = d[12], a31 = d[13], a32 = d[14], a33 = d[15];
      if (pp > 0) {                                                        // minus X X' with X = L(pp, pp - 1)
        const double* x = d - TS;
        double X[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) X[e] = x[e];
#define TZ_XX(i, j) (X[4 * i] * X[4 * j] + X[4 * i + 1] * X[4 * j + 1] + X[4 * i + 2] * X[4 * j + 2] + X[4 * i + 3] * X[4 * j + 3])
        a00 -= TZ_XX(0, 0); a10 -= TZ_XX(1, 0); a11 -= TZ_XX(1, 1); a20 -= TZ_XX(2, 0); a21 -= TZ_XX(2, 1); a22 -= TZ_XX(2, 2);
        a30 -= TZ_XX(3, 0); a31 -= TZ_XX(3, 1); a32 -= TZ_XX(3, 2); a33 -= TZ_XX(3, 3);
#undef TZ_XX
      }
      bool ok = true;
      double l00, i00, l11, i11, l22, i22, l33, i33;
      ok = ok && (a00 > 0.0);
      tz_sqrt_rsqrt(fmax(a00, 1e-300), l00, i00);
      const double l10 = a10 * i00, l20 = a20 * i00, l30 = a30 * i00;
      const double d1 = a11 - l10 * l10; ok = ok && (d1 > 0.0);
      tz_sqrt_rsqrt(fmax(d1, 1e-300), l11, i11);
      const double l21 = (a21 - l20 * l10) * i11, l31 = (a31 - l30 * l10) * i11;
      const double d2 = a22 - l20 * l20 - l21 * l21; ok = ok && (d2 > 0.0);
      tz_sqrt_rsqrt(fmax(d2, 1e-300), l22, i22);
      const double l32 = (a32 - l30 * l20 - l31 * l21) * i22;
      const double d3 = a33 - l30 * l30 - l31 * l31 - l32 * l32; ok = ok && (d3 > 0.0);
      tz_sqrt_rsqrt(fmax(d3, 1e-300), l33, i33);
      (void)l00; (void)l11; (void)l22; (void)l33;
      const double m10 = -l10 * i00 * i11;
      const double m21 = -l21 * i11 * i22;
      const double m32 = -l32 * i22 * i33;
      const double m20 = -(l20 * i00 + l21 * m10) * i22;
      const double m31 = -(l31 * i11 + l32 * m21) * i33;
      const double m30 = -(l30 * i00 + l31 * m10 + l32 * m20) * i33;
      if ((tz_tid() & 63) == 0) {
        if (!ok) *flag = 1;
        double* m = dinv + pp * 16;          // inverse of the diagonal factor (lower); the zeros above the diagonal are set once per launch
        m[0] = i00;
        m[4] = m10; m[5] = i11;
        m[8] = m20; m[9] = m21; m[10] = i22;
        m[12] = m30; m[13] = m31; m[14] = m32; m[15] = i33;
        dfac[0] = l10; dfac[1] = l20; dfac[2] = l21; dfac[3] = l30; dfac[4] = l31; dfac[5] = l32;
        dfac[6] = i00; dfac[7] = i11; dfac[8] = i22; dfac[9] = i33;
      }
      __builtin_amdgcn_s_setprio(0);
    } else if (pp > 0) {
#if TZ_PROFILE
      unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
      tz_tt_phase_a(Ht, TS, Tz, pp, wave - 1, 3);                               // (i) + (ii), see there
#if TZ_PROFILE
      unsigned long long tw1 = __builtin_amdgcn_s_memtime();
#endif
#if TZ_PROFILE
      if (p.prof && blockIdx.x == 0 && threadIdx.x == 64) { unsigned long long tw2 = __builtin_amdgcn_s_memtime(); atomicAdd(p.prof + 32, tw1 - tw0); atomicAdd(p.prof + 33, tw2 - tw1); }
#endif
    }
    if (pacc) { unsigned long long t1 = __builtin_amdgcn_s_memtime(); pacc[PH_CH_DIAG] += t1 - tc0; tc0 = t1; }
    __syncthreads();
    if (pacc) { unsigned long long t1 = __builtin_amdgcn_s_memtime(); pacc[PH_CH_UPD] += t1 - tc0; tc0 = t1; }
    {                                                                      // phase B: x L_pp' = a for every row below the diagonal tile
      const int t = tz_tid();
      const int nrow = 4 * (Tz - pp - 1);
      if (t < nrow) {
        const int I = pp + 1 + (t >> 2), i = t & 3;
        double* b = Ht + (tz_tri(I) + pp) * TS + 4 * i;
        const double b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
        const double l10 = dfac[0], l20 = dfac[1], l21 = dfac[2], l30 = dfac[3], l31 = dfac[4], l32 = dfac[5];
        const double x0 = b0 * dfac[6];
        const double x1 = (b1 - x0 * l10) * dfac[7];
        const double x2 = (b2 - x0 * l20 - x1 * l21) * dfac[8];
        const double x3 = (b3 - x0 * l30 - x1 * l31 - x2 * l32) * dfac[9];
        b[0] = x0; b[1] = x1; b[2] = x2; b[3] = x3;
      }
    }
    if (pacc) { unsigned long long t1 = __builtin_amdgcn_s_memtime(); pacc[PH_CH_PANEL] += t1 - tc0; tc0 = t1; }
    __syncthreads();
    if (pacc) { unsigned long long t1 = __builtin_amdgcn_s_memtime(); pacc[PH_CH_BAR] += t1 - tc0; tc0 = t1; }
  }
  return *flag == 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Block-diagonal inverses for the triangular solves.  Substitution tile by tile is a chain of Tz dependent steps of ~500 cycles
// each (cross-lane broadcast, 4x4 inverse, update); with the inverse W_Q = inv(L_QQ) of every 16 x 16 diagonal block the chain
// has Tz / 4 steps: y_Q = W_Q (r_Q - sum_{P<Q} L_QP y_P).  W_Q is built from the 4x4 tile inverses M_I the factorisation
// leaves in dinv:   W(I, I) = M_I,   W(I, J) = -M_I sum_{J <= K < I} L(I, K) W(K, J)   (tile rows I, J of the block),
// and overwrites the tiles of L_QQ (diagonal tiles included: the factorisation never stores them, the solves never read L there).
// One lane per column of W (16 lanes per block, four blocks per wave pass); all reads of a block precede its writes (lockstep).
// ---------------------------------------------------------------------------------------------------------------------------
__device__ inline void tz_tt_block_inverse(const IpmParams& p, double* Ht, const double* dinv) {
  const int Tz = p.Tz, TS = p.TS, nb = (Tz + 3) >> 2;
  const int t = tz_tid(), wave = t >> 6, lane = t & 63;
  for (int Q0 = 4 * wave; Q0 < nb; Q0 += 4 * TZ_NWAVES) {
    const int Q = Q0 + (lane >> 4), c = lane & 15, cj = c >> 2, ce = c & 3;     // column c of block Q: tile column cj, element ce
    const bool vq = Q < nb;
    const int R0 = 4 * (vq ? Q : 0);
    // tile row by tile row; the finished part of the column lives in LDS (tiles (K, cj), K < I: already W) and is read back, so
    // that only one tile row of operands is in registers at a time (the kernel is at its register budget here).  Lanes of a wave
    // run in lockstep: every read of tile row I precedes the writes of tile row I.
    for (int I = 0; I < 4; ++I) {
      const bool vi = vq && (R0 + I < Tz);
      const int ti = tz_tri(vi ? R0 + I : 0);
      double a0 = (I == cj && ce == 0) ? 1.0 : 0.0, a1 = (I == cj && ce == 1) ? 1.0 : 0.0;
      double a2 = (I == cj && ce == 2) ? 1.0 : 0.0, a3 = (I == cj && ce == 3) ? 1.0 : 0.0;      // e_c restricted to tile row I - sum_K L(I, K) w_K
      for (int K = cj; K < I; ++K) {                                            // w_K = 0 above the column's first tile row
        const double* wk = Ht + (tz_tri(vi ? R0 + K : 0) + R0 + cj) * TS + ce;     // W(4K + e, c), e = 0..3: stride 4
        const double w0 = wk[0], w1 = wk[4], w2 = wk[8], w3 = wk[12];
        const double* l = Ht + (ti + R0 + K) * TS;
        a0 -= (l[0] * w0 + l[1] * w1) + (l[2] * w2 + l[3] * w3);
        a1 -= (l[4] * w0 + l[5] * w1) + (l[6] * w2 + l[7] * w3);
        a2 -= (l[8] * w0 + l[9] * w1) + (l[10] * w2 + l[11] * w3);
        a3 -= (l[12] * w0 + l[13] * w1) + (l[14] * w2 + l[15] * w3);
      }
      const double* m = dinv + (vi ? R0 + I : 0) * 16;                           // M_I, lower triangular
      const double v0 = m[0] * a0;
      const double v1 = m[4] * a0 + m[5] * a1;
      const double v2 = (m[8] * a0 + m[9] * a1) + m[10] * a2;
      const double v3 = (m[12] * a0 + m[13] * a1) + (m[14] * a2 + m[15] * a3);
      tz_wave_sync();                                                            // all lanes have read tile row I
      if (vi && I >= cj) {
        double* o = Ht + (ti + R0 + cj) * TS + ce;
        o[0] = v0; o[4] = v1; o[8] = v2; o[12] = v3;
      }
      tz_wave_sync();                                                            // W of tile row I visible to the reads of the next rows
    }
  }
}

// (L L') out = rhs with the block inverses in place (tz_tt_block_inverse): thread t owns row t, a block of 16 rows is one DPP row
// of 16 lanes.  Block step Q, forward: the owners form y_Q = W_Q r_Q -- every lane needs the 16 residuals of its row of lanes:
// fifteen DPP row rotations against the pre-rotated row of W (wrot[k] = W[r][(r - k) mod 16], fetched ahead) -- and publish it in
// LDS; after one barrier every later row subtracts L(t, block Q) y_Q (16 FMAs, own row of L, y broadcast).  Backward: mirror image
// with the columns of W and of L.  ybuf: nzp doubles of LDS.
template <int K> struct TzRor { static __device__ inline double get(double v) { return tz_row_ror<K>(v); } };
template <> struct TzRor<0> { static __device__ inline double get(double v) { return v; } };

template <bool FWD>
__device__ inline double tz_tt_block_apply(const double (&wrot)[16], double rv) {
  // FWD: sum_k wrot[k] * rv[(lane - k) mod 16];  !FWD: sum_k wrot[k] * rv[(lane + k) mod 16]   (row_ror:N reads lane - N)
  double s0 = wrot[0] * rv, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#define TZ_BA(kk, acc) acc += wrot[kk] * (FWD ? TzRor<kk>::get(rv) : TzRor<16 - kk>::get(rv))
  TZ_BA(1, s1); TZ_BA(2, s2); TZ_BA(3, s3); TZ_BA(4, s0); TZ_BA(5, s1); TZ_BA(6, s2); TZ_BA(7, s3);
  TZ_BA(8, s0); TZ_BA(9, s1); TZ_BA(10, s2); TZ_BA(11, s3); TZ_BA(12, s0); TZ_BA(13, s1); TZ_BA(14, s2); TZ_BA(15, s3);
#undef TZ_BA
  return (s0 + s1) + (s2 + s3);
}

__device__ inline void tz_chol_solve_blk(const IpmParams& p, const double* Ht, const double* rhs, double* ybuf, double* out) {
  const int Tz = p.Tz, nzp = p.nzp, TS = p.TS, nb = (Tz + 3) >> 2;
  const int t = tz_tid(), jq = t & 3, tq = t >> 2, blkQ = t >> 4, r = t & 15;   // row t = 16 blkQ + r
  const bool live = t < nzp;
  double rv = live ? rhs[t] : 0.0;
  const double* rowp = Ht + tz_tri(live ? tq : 0) * TS + 4 * jq;                 // L(t, 4J + e) = rowp[J TS + e]
  // rows of L / columns of L' against one block of 16 unknowns (yy): the update every row outside the block takes
  auto row_dot = [&](int Q, const double* yy) {
    const double* lr = rowp + 4 * Q * TS;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      a0 += lr[J * TS] * yy[4 * J] + lr[J * TS + 2] * yy[4 * J + 2];
      a1 += lr[J * TS + 1] * yy[4 * J + 1] + lr[J * TS + 3] * yy[4 * J + 3];
    }
    return a0 + a1;
  };
  auto col_dot = [&](int Q, const double* xx) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      const int I = 4 * Q + J;
      if (I < Tz) {
        const double* lc = Ht + (tz_tri(I) + tq) * TS + jq;
        a0 += lc[0] * xx[4 * J] + lc[8] * xx[4 * J + 2];
        a1 += lc[4] * xx[4 * J + 1] + lc[12] * xx[4 * J + 3];
      }
    }
    return a0 + a1;
  };
  // ---- forward (one workgroup barrier per block: running the four blocks of a wave wave-synchronously, with a barrier per 64
  // rows only, measured 20 % slower -- the updates of the other waves' rows then wait for all four) -----------------------------
  {
    double wrot[16];                                                              // W[r][(r - k) mod 16]: zero above the diagonal (k > r)
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int c = (r - k) & 15; wrot[k] = (live && k <= r) ? rowp[(4 * blkQ + (c >> 2)) * TS + (c & 3)] : 0.0; }
    for (int Q = 0; Q < nb; ++Q) {
      const double y = tz_tt_block_apply<true>(wrot, rv);
      if (blkQ == Q && live) { rv = y; ybuf[t] = y; }
      if (Q + 1 < nb) {
        __syncthreads();
        if (blkQ > Q && live) rv -= row_dot(Q, ybuf + 16 * Q);
      }
    }
  }
  __syncthreads();                                                               // ybuf is reused by the backward sweep
  // ---- backward -------------------------------------------------------------------------------------------------------
  {
    double wrot[16];                                                              // W[(r + k) mod 16][r]: zero when r + k wraps (rows above) or beyond the matrix
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int rr = r + k, I = 4 * blkQ + (rr >> 2);
      wrot[k] = (live && rr < 16 && I < Tz) ? Ht[(tz_tri(I) + tq) * TS + 4 * (rr & 3) + jq] : 0.0;
    }
    for (int Q = nb - 1; Q >= 0; --Q) {
      const double x = tz_tt_block_apply<false>(wrot, rv);
      if (blkQ == Q && live) { rv = x; ybuf[t] = x; }
      if (Q > 0) {
        __syncthreads();
        if (blkQ < Q) rv -= col_dot(Q, ybuf + 16 * Q);                            // rows above take this block's x: column t of L(block Q, :)
      }
    }
  }
  if (live) out[t] = rv;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Solve (L L') out = rhs on the tile-triangle layout: thread t owns row t (nzp <= 256) in a register, wave w the 64-row block w.
// Forward: inside a block the 16 tile steps run wave-synchronously (the owner quad finishes its four unknowns with DPP
// broadcasts and the inverse diagonal tile, the other lanes pick them up by v_readlane); between blocks one barrier and one bulk
// update of the rows of later blocks through LDS.  Backward: the mirror image.  ybuf: nzp doubles of LDS.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ inline void tz_chol_solve_tt(const IpmParams& p, const double* Ht, const double* dinv, const double* rhs, double* ybuf, double* out) {
  const int Tz = p.Tz, nzp = p.nzp, TS = p.TS;
  const int t = tz_tid(), jq = t & 3, tq = t >> 2, wave = t >> 6, l0 = t & 63;
  const int nblk = (Tz + 15) >> 4;
  const bool live = t < nzp;
  double rv = live ? rhs[t] : 0.0;
  const double* rowp = Ht + tz_tri(live ? tq : 0) * TS + 4 * jq;             // + I TS + c : L(t, 4I + c)
  for (int blkI = 0; blkI < nblk; ++blkI) {                                 // ---- forward: L y = rhs
    const int I0 = 16 * blkI, I1 = min(Tz, I0 + 16);
    if (wave == blkI) {
      // operands of tile step I do not depend on the running solution: those of the next step are fetched while this one computes
      auto fload = [&](int I, double (&mm)[4], double (&qq)[4]) {
        const int Ic = I < I1 ? I : I1 - 1;
        const double* m = dinv + Ic * 16 + 4 * jq;                          // row jq of M_I = inv(L_II)
        const bool below = tq > Ic && live;
        const double* lr = rowp + Ic * TS;
#pragma unroll
        for (int e = 0; e < 4; ++e) { mm[e] = m[e]; qq[e] = below ? lr[e] : 0.0; }
      };
      auto fstep = [&](int I, const double (&mm)[4], const double (&qq)[4]) {
        const double a0 = tz_quad_bcast<0>(rv), a1 = tz_quad_bcast<1>(rv), a2 = tz_quad_bcast<2>(rv), a3 = tz_quad_bcast<3>(rv);
        const double yc = (mm[0] * a0 + mm[1] * a1) + (mm[2] * a2 + mm[3] * a3);     // y of this quad if it is the owner (tq == I)
        const int lb = 4 * (I - I0);
        const double y0 = tz_readlane(yc, lb), y1 = tz_readlane(yc, lb + 1), y2 = tz_readlane(yc, lb + 2), y3 = tz_readlane(yc, lb + 3);
        rv = (tq == I) ? yc : rv - ((qq[0] * y0 + qq[1] * y1) + (qq[2] * y2 + qq[3] * y3));
      };
      double mA[4], qA[4], mB[4], qB[4];
      fload(I0, mA, qA);
      for (int I = I0; I < I1; I += 2) {
        fload(I + 1, mB, qB);
        fstep(I, mA, qA);
        fload(I + 2, mA, qA);
        if (I + 1 < I1) fstep(I + 1, mB, qB);
      }
      if (live) ybuf[t] = rv;
    }
    if (blkI + 1 < nblk) {
      __syncthreads();
      if (wave > blkI && live) {                                            // bulk: rows of later blocks take this block's y
        double acc = 0.0;
        for (int I = I0; I < I1; ++I) {
          const double* lr = rowp + I * TS;
          const double* y = ybuf + 4 * I;
          acc += (lr[0] * y[0] + lr[1] * y[1]) + (lr[2] * y[2] + lr[3] * y[3]);
        }
        rv -= acc;
      }
    }
  }
  (void)l0;
  const int cofs = 4 * 0 + jq;                                              // L(4I + k, t) = Ht[(tri(I) + tq) TS + 4k + jq]
  for (int blkI = nblk - 1; blkI >= 0; --blkI) {                            // ---- backward: L' x = y
    const int I0 = 16 * blkI, I1 = min(Tz, I0 + 16);
    if (wave == blkI) {
      auto bload = [&](int I, double (&mm)[4], double (&qq)[4]) {
        const int Ic = I >= I0 ? I : I0;
        const double* m = dinv + Ic * 16 + jq;                              // column jq of M_I
        const bool above = tq < Ic && tq >= I0;
        const double* lc = Ht + (tz_tri(Ic) + (above ? tq : 0)) * TS + cofs;
#pragma unroll
        for (int e = 0; e < 4; ++e) { mm[e] = m[4 * e]; qq[e] = above ? lc[4 * e] : 0.0; }
      };
      auto bstep = [&](int I, const double (&mm)[4], const double (&qq)[4]) {
        const double a0 = tz_quad_bcast<0>(rv), a1 = tz_quad_bcast<1>(rv), a2 = tz_quad_bcast<2>(rv), a3 = tz_quad_bcast<3>(rv);
        const double xc = (mm[0] * a0 + mm[1] * a1) + (mm[2] * a2 + mm[3] * a3);
        const int lb = 4 * (I - I0);
        const double x0 = tz_readlane(xc, lb), x1 = tz_readlane(xc, lb + 1), x2 = tz_readlane(xc, lb + 2), x3 = tz_readlane(xc, lb + 3);
        rv = (tq == I) ? xc : rv - ((qq[0] * x0 + qq[1] * x1) + (qq[2] * x2 + qq[3] * x3));
      };
      double mA[4], qA[4], mB[4], qB[4];
      bload(I1 - 1, mA, qA);
      for (int I = I1 - 1; I >= I0; I -= 2) {
        bload(I - 1, mB, qB);
        bstep(I, mA, qA);
        bload(I - 2, mA, qA);
        if (I - 1 >= I0) bstep(I - 1, mB, qB);
      }
      if (live) ybuf[t] = rv;
    }
    if (blkI > 0) {
      __syncthreads();
      if (wave < blkI) {                                                    // bulk: rows of earlier blocks take this block's x
        double acc = 0.0;
        for (int I = I0; I < I1; ++I) {
          const double* lc = Ht + (tz_tri(I) + tq) * TS + cofs;
          const double* x = ybuf + 4 * I;
          acc += (lc[0] * x[0] + lc[4] * x[1]) + (lc[8] * x[2] + lc[12] * x[3]);
        }
        rv -= acc;
      }
    }
  }
  if (live) out[t] = rv;
}
