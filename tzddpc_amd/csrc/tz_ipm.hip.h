// Batched Mehrotra predictor-corrector interior point kernel: one 256-thread workgroup per trajectory.
//
//   min 1/2 x'Px + q'x   s.t.  G x + s = h,  s >= 0          (scaled, one-sided; lambda >= 0)
//
// per iteration:  H = P + G' diag(lambda/s) G + reg I   (v_mfma_f64_4x4x4, block-sparse plan, operands prefetched 3 deep)
//                 H = L L'                               (tile-4 blocked Cholesky in LDS, trailing updates by MFMA)
//                 two solves (predictor, corrector), five G / G' mat-vecs, step-length reductions.
// Row quantities (s, lambda, ds, dlambda, h, Gx, ...) live in registers of the thread that owns the row
// (MAXR rows per thread); LDS holds H (lower triangle, quads in MFMA lane order), the nz-vectors and one
// row-vector staging buffer, so that four workgroups fit on a CU for the benchmark sizes.
#pragma once

#define TZ_MINWAVES 4
#define TZ_RLEV_CARRY_MAX 2      // diagonal-shift level (1e-6) a warm-started step may inherit from the step before ...
#ifndef TZ_RLEV_GATE
#define TZ_RLEV_GATE 1e6         // ... from the iteration on whose complementarity is within this factor of the one at which that step broke down
#endif
#define TZ_SEED_VIOL_MAX 0.1     // stored start: largest violation of the new rows (equilibrated) it is still used at
// H lives in LDS as tile rows of quads (4 column tiles); a quad is 4 matrix rows of 16 doubles padded to TZ_QROW = 17 so
// that neither the MFMA accumulator access (row-major inside the quad) nor the column access of the factorisation and the
// triangular solves runs into LDS bank conflicts (row stride 16 doubles = 32 banks collides 4- to 8-way).
#define TZ_QROW 17
#define TZ_QSTR (4 * TZ_QROW)
struct IpmItem { int I0, q0, nq, kptr, klen; };

// G in balanced "lane-ELL" form for the matrix-vector products (built by the host, tz_problem_create): the non-zeros of every
// output (a row of G for G x, a column for G'v) are dealt to as many consecutive virtual lanes as it takes to give every lane
// at most L entries; entry e of virtual lane v is at [(pass * L + e) * NL + lane] with v = pass * NL + lane (NL = 256 for G x,
// 192 for G'v: waves 1-3, wave 0 does something else meanwhile).  Every lane walks L entries -- coalesced, no zeros fetched,
// equal work per lane -- and leaves one partial sum in LDS; the owner of an output adds its lanes' partial sums in a fixed
// order (seg[o] = first lane | lanes << 16).  An entry is ONE 16-byte record (value, byte offset of the input entry, pad): one
// vector-memory instruction and one address add per non-zero (the kernel is issue-bound, not byte-bound: 10-byte (value, 16-bit
// index) pairs in two arrays cost two loads and a shift-add; a 4-byte value dictionary cost three LDS gathers and was slower still).
// The problems with more than 64 variables (tile-triangle class) keep the COMPACT form instead -- 8-byte values and 16-bit indices in two
// arrays, 10 bytes per non-zero: their products are paced by the rate at which G arrives from L2 (1-2 MB per pass, six passes per
// iteration), and 16-byte records cost them 10 % (DI N = 80) to 14 % (5-dim, two inputs) of the kernel time (profiles/r4w_ab_r3_vs_r4.txt).
struct TzEll { int L, VL; const TzEllEnt* ent; const int* seg; const double* val; const unsigned short* idx; };      // TzEllEnt, tz_d2, tz_ell_off: tz_kernels.hip.h
// one virtual lane's walk over its L records, NL lanes apart.  base: wave-uniform address of record 0 of lane 0 of the pass (scalar
// registers, advanced by scalar adds); lo: this lane's byte offset (one 32-bit vector register): no vector address arithmetic per load
typedef __attribute__((address_space(1))) const char* tz_gptr;
typedef __attribute__((address_space(1))) const tz_d2* tz_gd2ptr;
__device__ inline double tz_ell_walk(const char* base_, unsigned lo, int L, unsigned stride, const double* in) {
  double a0 = 0.0, a1 = 0.0;
  const char* b = reinterpret_cast<const char*>(in);
  tz_gptr base = (tz_gptr)base_;
  int e = 0;
  for (; e + 3 < L; e += 4) {
    // the four record addresses as scalar base + 32-bit lane offset (the bases pinned in scalar registers: the compiler otherwise folds
    // base + lane offset into ONE 64-bit vector address and derives the other three from it with vector adds)
    tz_gptr q0 = base, q1 = base + stride, q2 = base + 2 * (size_t)stride, q3 = base + 3 * (size_t)stride;
    asm volatile("" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3));
    const tz_d2 r0 = *(tz_gd2ptr)(q0 + lo), r1 = *(tz_gd2ptr)(q1 + lo), r2 = *(tz_gd2ptr)(q2 + lo), r3 = *(tz_gd2ptr)(q3 + lo);
    base += 4 * (size_t)stride;
    a0 += r0.x * *reinterpret_cast<const double*>(b + tz_ell_off(r0.y));
    a1 += r1.x * *reinterpret_cast<const double*>(b + tz_ell_off(r1.y));
    a0 += r2.x * *reinterpret_cast<const double*>(b + tz_ell_off(r2.y));
    a1 += r3.x * *reinterpret_cast<const double*>(b + tz_ell_off(r3.y));
  }
  for (; e < L; ++e) {
    tz_gptr q0 = base;
    asm volatile("" : "+s"(q0));
    const tz_d2 r = *(tz_gd2ptr)(q0 + lo);
    base += stride;
    a0 += r.x * *reinterpret_cast<const double*>(b + tz_ell_off(r.y));
  }
  return a0 + a1;
}

// compact form: entry e of the lane at val[e * NL], idx[e * NL] (the caller offsets the pointers to the lane)
__device__ inline double tz_ell_walk_c(const double* val, const unsigned short* idx, int L, int NL, const double* in) {
  double a0 = 0.0, a1 = 0.0;
  int e = 0;
  for (; e + 3 < L; e += 4) {
    const double x0 = val[(size_t)e * NL], x1 = val[(size_t)(e + 1) * NL], x2 = val[(size_t)(e + 2) * NL], x3 = val[(size_t)(e + 3) * NL];
    const int i0 = idx[(size_t)e * NL], i1 = idx[(size_t)(e + 1) * NL], i2 = idx[(size_t)(e + 2) * NL], i3 = idx[(size_t)(e + 3) * NL];
    a0 += x0 * in[i0]; a1 += x1 * in[i1]; a0 += x2 * in[i2]; a1 += x3 * in[i3];
  }
  for (; e < L; ++e) a0 += val[(size_t)e * NL] * in[idx[(size_t)e * NL]];
  return a0 + a1;
}

struct IpmParams {
  int B, nz, mi, nzp, mip, Tz, Kc, nquads, nklist, nP;   // nP: rows of P beyond which P is zero
  const double* P;       // nzp x nzp
  const double* Gp;      // (Kc+1) x (Tz+1) x 16 patches: Gp[(kc*(Tz+1) + J)*16 + 4k + j] = G[4kc+k][4J+j]; tile Tz of every row and row Kc are zero
  const IpmItem* items;  // Gram work items, grouped per wave
  const int* item_ptr;   // TZ_NWAVES + 1
  const int* klist;
  TzEll eg, et;          // G x (outputs = rows) and G'v (outputs = columns)
  const int* smask;      // ksplit: per super-step (16 rows of G) bit J set when tile column J has a non-zero; smask[S] = 0
  const double* q; const double* h;
  const int* prestatus;
  double* x; double* s; double* lam;
  int* status; int* iters;
  const int* prev_status;   // warm start gate: status of the previous closed-loop step (may alias nothing else); null = cold
  int* status_copy;         // optional second destination of the status (library-owned copy for the next step)
  int max_iter; double tol, reg, step_frac;
  double inv_mi;              // 1 / mi
  double mu_floor, tol_loose, step_frac_retry;   // 1e-3 mu_tol, 1e3 tol, min(step_frac, 0.99): formed on the host (uniform f64 expressions
                              // have no scalar ALU: the compiler hoists them out of the step loop into vector registers and spills them)
  double mu_tol;              // complementarity target (<= tol): the distance to the solution of a degenerate problem goes like sqrt(mu)
  double tol_res;             // residual tolerance of the stopping test (>= tol: what the reference's conic solvers ask of feasibility)
  int warm_steps;             // multi-step launches: step k+1 starts from the solution of step k
  int nell;                   // max(eg.VL, et.VL)
  int ntube;                  // doubles of tube tables kept in LDS by fused launches: (pmax + 1) n n + pmax (n + m) n
  int shift_policy;           // receding-horizon shift of the warm start: 0 never, 1 always, k >= 2 after a step of >= k iterations
                              // and for as long as the shifted steps that follow finish in one iteration
  int* shift_state;           // per trajectory: 0 = the last warm start was not shifted, 1 + q = it was, after q quiet (one-iteration) shifted steps
  int shift_quiet;            // k >= 2 policies: leave the shifted regime after this many quiet steps in a row (0: never)
  const int* sx; const int* sr; const double* sxs; const double* sls;   // source variable / row and rescaling, see tz_problem_desc
  // problems with more than 64 variables (tz_tt.hip.h): H in the tile-triangle layout, TS doubles per tile, blocked Gram plan
  int TS, ntile, gu;       // gu: tiles per side of a unit of the blocked Gram
  const struct TzGUnit* gunits; const int* gunit_ptr;      // units of the blocked Gram, grouped per wave (TZ_NWAVES + 1 offsets)
  int chol1;                  // Tz <= 16: one wave factors H while the other three form the predictor's right-hand side
  int ksplit;                 // Gram by tz_form_H_ksplit (Tz <= TZ_KS_TZ) instead of the item plan
  int warm; double warm_floor;   // warm != 0: start from the x / lambda already stored for the trajectory (closed-loop steps)
  double warm_gain, warm_cap;    // warm point pushed into the cone by sigma = min(max(warm_floor, warm_gain * violation of the new rows), warm_cap)
  double aff_thr, aff_mu;        // predictor step taken as the step (no corrector solve) when it reaches aff_thr of the way to the
                                 // boundary un-damped and leaves mu_aff <= aff_mu * mu; aff_thr > 1 disables
  unsigned long long* work;   // [0] += factorisations, [1] += trajectory solves, [2] = max over trajectories of the factorisations of one launch; may be null
  unsigned long long* prof;   // TZ_PROF=1: per-phase cycle sums of workgroup 0 (diagnostic; no output depends on it)
  FuseParams F;               // F.on != 0: whole closed-loop step in this launch (q, h, prestatus above are then unused)
};

enum { PH_FORM = 0, PH_CHOL = 1, PH_SOLVE = 2, PH_GEMVT = 3, PH_GEMV = 4, PH_ELEM = 5, PH_TOTAL = 6, PH_CH_UPD = 8, PH_CH_DIAG = 9, PH_CH_PANEL = 10, PH_CH_BAR = 11, PH_PROLOGUE = 12, PH_EPILOGUE = 13, PH_GRAM_LOOP = 14, PH_GRAM_RED = 15, PH_GRAM_BAR = 16, PH_GRAM_RMW = 17, PH_TUBE = 18, PH_WARM = 19, PH_TOP = 20, PH_STEP = 21,
       PH_RD_A = 22, PH_RD_B = 23, PH_RD_C = 24, PH_EPI_A = 25, PH_EPI_B = 26, PH_MAPS_Q = 27, PH_WARM_A = 28, PH_WARM_B = 29, PH_TEST = 30, PH_T1 = 31, PH_T2 = 32, PH_T3 = 33, PH_COUNT = 34 };   // 22 ..: finer stamps of the fixed part (diagnostic build)

__device__ inline int tz_qprefix(int I) {   // number of quads in tile rows < I (row I has (I>>2)+1 quads)
  int a = I >> 2, b = I & 3;
  return (a + 1) * (2 * a + b);
}
__device__ inline int tz_hidx(int r, int c) {   // LDS index of H(r, c), r >= c (tile-row major, quads in lane order)
  int I = r >> 2, J = c >> 2;
  return (tz_qprefix(I) + (J >> 2)) * TZ_QSTR + TZ_QROW * (r & 3) + 4 * (J & 3) + (c & 3);
}

// threadIdx.x behind an optimisation barrier.  Every routine below derives its lane offsets from a fresh copy: otherwise the
// compiler computes the per-lane address of every array once at kernel entry (they are loop invariant), runs out of registers,
// spills them, and re-loads them from scratch -- a memory round trip in place of one integer add.
__device__ inline int tz_tid() {
  int v = threadIdx.x;
  asm volatile("" : "+v"(v));
  return v;
}

// s_setprio level of the wave that runs a serial stretch of its workgroup (factorisation, triangular solves): four workgroups share
// a CU and their waves compete for the same SIMD issue slots; putting the wave the other three are waiting for first is worth
// 3.7 % on the bench problem (2 + 2).  The forward substitution that trails the factorisation is not on the critical path (0).
#define TZ_PRIO 3
#define TZ_PRIO_SOLVE 3
// The same idea by phase: the short barrier-separated phases between two solves (stopping test passed -> recovery, plant update, tube,
// maps, warm start) at 2, the element-wise / reduction phases of an iteration at 1, the bulk phases (Gram, G products) at 0:
// another 3.8 % (A/B on one box: 2 / 1 beats 1 / 0, 1 / 1 and 2 / 2).
#define TZ_PRIO_ELEM 1
#define TZ_PRIO_GLUE 2
enum { RED_SUM = 0, RED_MAX = 1, RED_MIN = 2, RED_MAXU = 3 };

// 64-bit DPP move (two 32-bit halves).  mov_dpp with bound_ctrl instead of update_dpp(0, ...): no `old` operand to initialise -- the
// compiler materialised the zero with two more moves per call (quad permutes and row rotations read valid lanes only: same values)
template <int CTRL>
__device__ inline double tz_dpp_mov(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// (fmax / fmin put a canonicalising v_max_f64 x, x in front of operands not known to be quiet; the bare instruction through inline
// assembly removes it -- and costs 36 B/lane more scratch in the kernel's register allocation: 2 % slower overall.  Left as fmax.)
__device__ inline double tz_readlane(double v, int lane) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int OP>
__device__ inline double tz_op(double a, double b) { return (OP == RED_SUM) ? a + b : ((OP == RED_MAX || OP == RED_MAXU) ? fmax(a, b) : fmin(a, b)); }

// Reduction over the 64 lanes of a wave, result in every lane.  Cross-lane moves by DPP (quad permutes, row rotations) and four
// v_readlane for the rows of 16 -- no ds_bpermute round trips through the LDS crossbar.  Fixed combination order.
// DPP move that only writes the rows of 16 lanes selected by ROWMASK (the others keep `old`)
template <int CTRL, int ROWMASK>
__device__ inline double tz_dpp_mov_rows(double old, double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v), o = __builtin_bit_cast(unsigned long long, old);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)u, CTRL, ROWMASK, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(u >> 32), CTRL, ROWMASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int OP>
__device__ inline double tz_wave_reduce(double v) {
  v = tz_op<OP>(v, tz_dpp_mov<0xB1>(v));             // quad_perm [1,0,3,2]
  v = tz_op<OP>(v, tz_dpp_mov<0x4E>(v));             // quad_perm [2,3,0,1]
  v = tz_op<OP>(v, tz_dpp_mov<0x124>(v));            // row_ror:4
  v = tz_op<OP>(v, tz_dpp_mov<0x128>(v));            // row_ror:8  -> every lane holds the result of its row of 16
  // rows combined by the two row broadcasts (lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2, 3): lane 63 ends
  // with the result of the wave, which leaves through two v_readlane -- fewer instructions than reading the four rows out
  const double idn = (OP == RED_SUM) ? 0.0 : v;      // what the rows that receive nothing combine with
  v = tz_op<OP>(v, tz_dpp_mov_rows<0x142, 0xA>(idn, v));        // row_bcast:15
  v = tz_op<OP>(v, tz_dpp_mov_rows<0x143, 0xC>((OP == RED_SUM) ? 0.0 : v, v));   // row_bcast:31
  return tz_readlane(v, 63);
}

// up to three simultaneous block reductions (ops fixed at compile time; the first NV of a, b, c take part); result broadcast to
// all threads.  The exchange buffer alternates between red[0..11] and red[16..27] (`par`, toggled here: every thread makes the
// same sequence of calls), so one barrier per reduction is enough: a buffer is written again only after the barrier of the
// reduction in between, which no thread reaches before it has read its values of this one.
// up to three values through the stages of tz_wave_reduce side by side (the compiler keeps three calls one after the other: three
// dependent chains of six stages in sequence instead of one chain of six stages three wide)
template <int OP0, int OP1, int OP2, int NV>
__device__ inline void tz_wave_reduce3(double& a, double& b, double& c) {
#define TZ_STAGE(CTRL) do { const double ta = tz_dpp_mov<CTRL>(a), tb = (NV > 1) ? tz_dpp_mov<CTRL>(b) : 0.0, tc = (NV > 2) ? tz_dpp_mov<CTRL>(c) : 0.0; \
    a = tz_op<OP0>(a, ta); if (NV > 1) b = tz_op<OP1>(b, tb); if (NV > 2) c = tz_op<OP2>(c, tc); } while (0)
  TZ_STAGE(0xB1); TZ_STAGE(0x4E); TZ_STAGE(0x124); TZ_STAGE(0x128);
#undef TZ_STAGE
  {
    const double ta = tz_dpp_mov_rows<0x142, 0xA>((OP0 == RED_SUM) ? 0.0 : a, a), tb = (NV > 1) ? tz_dpp_mov_rows<0x142, 0xA>((OP1 == RED_SUM) ? 0.0 : b, b) : 0.0,
                 tc = (NV > 2) ? tz_dpp_mov_rows<0x142, 0xA>((OP2 == RED_SUM) ? 0.0 : c, c) : 0.0;
    a = tz_op<OP0>(a, ta); if (NV > 1) b = tz_op<OP1>(b, tb); if (NV > 2) c = tz_op<OP2>(c, tc);
  }
  {
    const double ta = tz_dpp_mov_rows<0x143, 0xC>((OP0 == RED_SUM) ? 0.0 : a, a), tb = (NV > 1) ? tz_dpp_mov_rows<0x143, 0xC>((OP1 == RED_SUM) ? 0.0 : b, b) : 0.0,
                 tc = (NV > 2) ? tz_dpp_mov_rows<0x143, 0xC>((OP2 == RED_SUM) ? 0.0 : c, c) : 0.0;
    a = tz_op<OP0>(a, ta); if (NV > 1) b = tz_op<OP1>(b, tb); if (NV > 2) c = tz_op<OP2>(c, tc);
  }
  a = tz_readlane(a, 63); if (NV > 1) b = tz_readlane(b, 63); if (NV > 2) c = tz_readlane(c, 63);
}

// RED_MAXU: maximum of NON-NEGATIVE values, rounded UP to the next multiple of 2^-20 relative (or left exact when it is inf / NaN).
// The order of non-negative doubles is the order of their bit patterns, so the maximum of the HIGH 32-bit words is found with one
// v_max_u32 per stage, the cross-lane move fused into it as its DPP operand -- 7 vector instructions per value instead of the 19 of
// the f64 tree (every f64 stage is two 32-bit DPP moves plus the operation: f64 instructions take no DPP operand).  Every use is a
// quantity that only enters a threshold test or the length of a step (residual norms, step-to-boundary ratios, the violation that
// sizes the warm-start push, the scales of the stopping test): overestimating it by <= 1e-6 relative makes the test a hair stricter
// and the step a hair shorter.  Deterministic; NaN and inf survive (their patterns are the largest).
template <int CTRL, int ROWMASK>
__device__ inline unsigned tz_dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, ROWMASK == 0xf); }
__device__ inline unsigned tz_hi(double v) { return (unsigned)(__builtin_bit_cast(unsigned long long, v) >> 32); }
__device__ inline unsigned tz_wave_maxu(unsigned u) {
  u = max(u, tz_dpp_u32<0xB1, 0xf>(u)); u = max(u, tz_dpp_u32<0x4E, 0xf>(u));
  u = max(u, tz_dpp_u32<0x124, 0xf>(u)); u = max(u, tz_dpp_u32<0x128, 0xf>(u));
  u = max(u, tz_dpp_u32<0x142, 0xA>(u)); u = max(u, tz_dpp_u32<0x143, 0xC>(u));
  return (unsigned)__builtin_amdgcn_readlane((int)u, 63);
}
__device__ inline double tz_from_hi_up(unsigned u) {
  u += (u < 0x7FEFFFFFu) ? 1u : 0u;
  return __builtin_bit_cast(double, (unsigned long long)u << 32);
}

// the four waves' results of one value (slots q[0..3]); a RED_MAXU value is a high word parked in the low half of its slot
template <int OP>
__device__ inline double tz_fold4(const double* q) {
  if (OP == RED_MAXU) {
    const unsigned* u = reinterpret_cast<const unsigned*>(q);
    return tz_from_hi_up(max(max(u[0], u[2]), max(u[4], u[6])));
  }
  return tz_op<OP>(tz_op<OP>(tz_op<OP>(q[0], q[1]), q[2]), q[3]);
}

template <int OP0, int OP1, int OP2, int NV = 3>
__device__ inline void tz_block_reduce3(double& a, double& b, double& c, double* red, int& par) {
  constexpr bool U0 = OP0 == RED_MAXU, U1 = NV > 1 && OP1 == RED_MAXU, U2 = NV > 2 && OP2 == RED_MAXU;
  // the f64 values go through the stages side by side; the RED_MAXU ones as high words
  unsigned ua = 0, ub = 0, uc = 0;
  if (U0) ua = tz_wave_maxu(tz_hi(a));
  if (U1) ub = tz_wave_maxu(tz_hi(b));
  if (U2) uc = tz_wave_maxu(tz_hi(c));
  constexpr int NF = (U0 ? 0 : 1) + ((NV > 1 && !U1) ? 1 : 0) + ((NV > 2 && !U2) ? 1 : 0);
  if (NF == 3) tz_wave_reduce3<OP0, OP1, OP2, 3>(a, b, c);
  else if (NF == 2) { if (U0) tz_wave_reduce3<OP1, OP2, OP2, 2>(b, c, a); else if (U1) tz_wave_reduce3<OP0, OP2, OP2, 2>(a, c, b); else tz_wave_reduce3<OP0, OP1, OP1, 2>(a, b, c); }
  else if (NF == 1) { if (!U0) tz_wave_reduce3<OP0, OP0, OP0, 1>(a, b, c); else if (NV > 1 && !U1) tz_wave_reduce3<OP1, OP1, OP1, 1>(b, a, c); else tz_wave_reduce3<OP2, OP2, OP2, 1>(c, a, b); }
  const int tt = tz_tid();
  int lane = tt & 63, w = tt >> 6;
  double* rb_ = red + (par ? 16 : 0);
  par ^= 1;
  if (lane == 0) {
    rb_[w] = U0 ? __builtin_bit_cast(double, (unsigned long long)ua) : a;
    if (NV > 1) rb_[4 + w] = U1 ? __builtin_bit_cast(double, (unsigned long long)ub) : b;
    if (NV > 2) rb_[8 + w] = U2 ? __builtin_bit_cast(double, (unsigned long long)uc) : c;
  }
  __syncthreads();
  const double ra = tz_fold4<OP0>(rb_), rb = (NV > 1) ? tz_fold4<OP1>(rb_ + 4) : 0.0, rc = (NV > 2) ? tz_fold4<OP2>(rb_ + 8) : 0.0;
  a = ra; if (NV > 1) b = rb; if (NV > 2) c = rc;
}

// part[w][c] = sum over the rows r = w, w+4, ... of M[r][c] in[r]  (row-major M, rows x nzp; `in` in LDS).
// NCG = ceil(nzp / 64) column groups per lane.  The four per-wave partials are combined by tz_gemvT_get.
// W0 / NW: the product is shared by the NW waves W0 .. W0 + NW - 1 (the others must not call); partial sums part[0 .. NW).
template <int NCG, int W0 = 0, int NW = TZ_NWAVES>
__device__ inline void tz_gemvT_partial(const double* M, int rows, int nzp, const double* in, double* part) {
  const int tt = tz_tid();
  const int lane = tt & 63, w = (tt >> 6) - W0;
  constexpr int UR = (NCG <= 2) ? 8 : 4;
  double acc[NCG];
#pragma unroll
  for (int g = 0; g < NCG; ++g) acc[g] = 0.0;
  int r = w;
  for (; r + (UR - 1) * NW < rows; r += UR * NW) {
    double v[UR], m[UR][NCG];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      v[u] = in[r + u * NW];
      const double* row = M + (size_t)(r + u * NW) * nzp;
#pragma unroll
      for (int g = 0; g < NCG; ++g) { const int c = min(lane + 64 * g, nzp - 1); m[u][g] = row[c]; }   // clamped: columns >= nzp are never stored
    }
#pragma unroll
    for (int u = 0; u < UR; ++u)
#pragma unroll
      for (int g = 0; g < NCG; ++g) acc[g] += m[u][g] * v[u];
  }
  for (; r < rows; r += NW) {
    const double v = in[r];
    const double* row = M + (size_t)r * nzp;
#pragma unroll
    for (int g = 0; g < NCG; ++g) { const int c = min(lane + 64 * g, nzp - 1); acc[g] += row[c] * v; }
  }
#pragma unroll
  for (int g = 0; g < NCG; ++g) { const int c = lane + 64 * g; if (c < nzp) part[w * nzp + c] = acc[g]; }
}
__device__ inline double tz_gemvT_get(const double* part, int nzp, int c) {
  return (part[c] + part[nzp + c]) + (part[2 * nzp + c] + part[3 * nzp + c]);
}
__device__ inline double tz_gemvT_get3(const double* part, int nzp, int c) {
  return (part[c] + part[nzp + c]) + part[2 * nzp + c];
}

// out[k] = (G in)_r for the rows r = t + 256 k this thread owns; `in` (nz entries) and pl (eg.VL doubles) in LDS.  Two halves: every
// thread walks its virtual lanes (tz_ell_gemv_walk; pl must not be in use by the owners of an earlier product), and after a workgroup
// barrier the owner of a row adds its lanes' partial sums (tz_ell_gemv_sum).
template <bool COMPACT>
__device__ inline void tz_ell_gemv_walk(const IpmParams& p, const double* in, double* pl) {
  const int t = tz_tid(), L = p.eg.L;
  for (int v0 = 0; v0 < p.eg.VL; v0 += TZ_THREADS) {
    if constexpr (COMPACT) pl[v0 + t] = tz_ell_walk_c(p.eg.val + (size_t)v0 * L + t, p.eg.idx + (size_t)v0 * L + t, L, TZ_THREADS, in);
    else pl[v0 + t] = tz_ell_walk(reinterpret_cast<const char*>(p.eg.ent + (size_t)v0 * L), (unsigned)t * 16u, L, TZ_THREADS * 16u, in);
  }
}
template <int MAXR>
__device__ inline void tz_ell_gemv_sum(const double* pl, const int (&rseg)[MAXR], double (&out)[MAXR]) {
#pragma unroll
  for (int k = 0; k < MAXR; ++k) {
    double a = 0.0;
    int sg = rseg[k];
    asm volatile("" : "+v"(sg));                     // launch-invariant per lane: keep the derived addresses out of the hoisted (and spilled) set
    const int first = sg & 0xffff, cnt = sg >> 16;
    for (int j = 0; j < cnt; ++j) a += pl[first + j];
    out[k] = a;
  }
}
template <int MAXR, bool COMPACT>
__device__ inline void tz_ell_gemv(const IpmParams& p, const double* in, double* pl, const int (&rseg)[MAXR], double (&out)[MAXR]) {
  __syncthreads();                                   // pl may still be read by the owners of the previous product
  tz_ell_gemv_walk<COMPACT>(p, in, pl);
  __syncthreads();
  tz_ell_gemv_sum<MAXR>(pl, rseg, out);
}

// Partial sums of G'in by the 192 threads of waves 1-3 (the caller keeps wave 0 out); `in` (mi entries), pl (et.VL doubles) in
// LDS.  After the next workgroup barrier tz_ell_colsum(pl, cseg) is column c's value for the thread holding cseg = et.seg[c].
template <bool COMPACT>
__device__ inline void tz_ell_gemvT_part(const IpmParams& p, const double* in, double* pl) {
  constexpr int NL = TZ_THREADS - 64;
  const int l = tz_tid() - 64, L = p.et.L;
  for (int v0 = 0; v0 < p.et.VL; v0 += NL) {
    if constexpr (COMPACT) pl[v0 + l] = tz_ell_walk_c(p.et.val + (size_t)v0 * L + l, p.et.idx + (size_t)v0 * L + l, L, NL, in);
    else pl[v0 + l] = tz_ell_walk(reinterpret_cast<const char*>(p.et.ent + (size_t)v0 * L), (unsigned)l * 16u, L, NL * 16u, in);
  }
}
__device__ inline double tz_ell_colsum(const double* pl, int cseg) {
  double a = 0.0;
  asm volatile("" : "+v"(cseg));                     // as in tz_ell_gemv
  const int first = cseg & 0xffff, cnt = cseg >> 16;
  for (int j = 0; j < cnt; ++j) a += pl[first + j];
  return a;
}

// Gram matrix  H = P + G' diag(w) G + reg I  into LDS quads, by v_mfma_f64_4x4x4 (blk = 4 column tiles).
// wv: LDS, mip + 4 entries, entries >= mi are zero.  kl: k-lists in LDS.
struct TzStage { double a[4]; double b[2]; double w; };

// A load the optimiser may neither duplicate nor sink to its use: a relaxed wavefront-scope atomic load is an ordinary
// global_load in the ISA, but unlike a plain load from read-only memory it cannot be rematerialised, which is what keeps
// the operand prefetch of the Gram loop several stages ahead of the MFMAs that consume it.
__device__ inline double tz_ld_pinned(const double* p) {
  unsigned long long u = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  return __builtin_bit_cast(double, u);
}

__device__ inline void tz_form_H(const IpmParams& p, double* Hq, const double* wv, const int* kl) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
  const int Tz = p.Tz;
  // Patch row kc = (Tz + 1) tiles of 16 doubles; tile Tz is all zero (target of masked operands).  The row base is wave
  // uniform (scalar registers), every operand is  row base + 32-bit lane offset:  no vector address arithmetic per load.
  const unsigned rowbytes = (unsigned)(Tz + 1) * 128u;
  const unsigned laneoff = (unsigned)(4 * k + ij) * 8u;
  const char* gp = (const char*)p.Gp;
  const int it0 = __builtin_amdgcn_readfirstlane(p.item_ptr[wave]), it1 = __builtin_amdgcn_readfirstlane(p.item_ptr[wave + 1]);
  for (int it = it0; it < it1; ++it) {
    const IpmItem item = p.items[it];
    const int I0 = __builtin_amdgcn_readfirstlane(item.I0), q0 = __builtin_amdgcn_readfirstlane(item.q0);
    const int nq = __builtin_amdgcn_readfirstlane(item.nq), klen = __builtin_amdgcn_readfirstlane(item.klen);
    const int* kli = kl + __builtin_amdgcn_readfirstlane(item.kptr);
    double acc[4][2];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) { acc[ii][0] = 0.0; acc[ii][1] = 0.0; }
    unsigned aoff[4], boff[2];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) aoff[ii] = (unsigned)((I0 + ii < Tz) ? (I0 + ii) : Tz) * 128u + laneoff;
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) { const int J = 4 * (q0 + nn) + blk; boff[nn] = (unsigned)((nn < nq && J < Tz) ? J : Tz) * 128u + laneoff; }
    auto load = [&](int kk, TzStage& st) {
      const int kc = __builtin_amdgcn_readfirstlane(kli[kk]);            // lists are padded with Kc (all-zero patch row)
      st.w = wv[4 * kc + k];
      const char* prow = gp + (size_t)kc * rowbytes;                      // scalar
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) st.a[ii] = tz_ld_pinned((const double*)(prow + aoff[ii]));
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) st.b[nn] = tz_ld_pinned((const double*)(prow + boff[nn]));
    };
    auto fma8 = [&](const TzStage& st) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const double a = st.a[ii] * st.w;
        acc[ii][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, st.b[0], acc[ii][0], 0, 0, 0);
        acc[ii][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, st.b[1], acc[ii][1], 0, 0, 0);
      }
    };
    TzStage s0, s1, s2, s3;
    load(0, s0); load(1, s1); load(2, s2);
    for (int kk = 0; kk < klen; kk += 4) {
      load(kk + 3, s3); fma8(s0);
      load(kk + 4, s0); fma8(s1);
      load(kk + 5, s1); fma8(s2);
      load(kk + 6, s2); fma8(s3);
    }
    // D lane (i = lane>>4, blk, j = lane&3) = H(4I + i, 4(4q + blk) + j)
    const int i = lane >> 4, j = lane & 3;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int I = I0 + ii;
      if (I >= Tz) continue;
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        const int q = q0 + nn;
        if (nn >= nq || q > (I >> 2)) continue;
        const int r = 4 * I + i, c = 4 * (4 * q + blk) + j;
        double v = acc[ii][nn];
        if (c < p.nzp) v += p.P[(size_t)r * p.nzp + c];
        if (r == c) v = (r < p.nz) ? v + p.reg : 1.0;
        Hq[(tz_qprefix(I) + q) * TZ_QSTR + TZ_QROW * i + 4 * blk + j] = v;
      }
    }
  }
}

// Gram matrix for small problems (Tz <= TZ_KS_TZ tile columns, i.e. nz <= 40).
// The four blocks of v_mfma_f64_4x4x4 take four DIFFERENT patch rows (blk = patch row 4s + blk of "super-step" s) of the
// SAME output tile (I, J): lane (k, blk, ij) loads element [k][ij] of every tile of its patch row once -- 64 distinct
// doubles per load instruction, nothing is fetched twice into registers (the L1 -> register path, 64 B/clk per CU, is what
// bounds every pass over G) -- and that one value is the A operand (times w) of the tiles in row I and the B operand of
// the tiles in column J.  One MFMA per (super-step, tile); tiles whose column I or J is entirely zero in the super-step
// are skipped (wave-uniform bit tests of a host-built mask).  The waves split the work 2 x 2: wave>>1 picks the tile-row
// range [R0, R1), wave&1 the even / odd super-steps; the partial sums are folded over blk with DPP row rotations and the
// two halves added through LDS in a fixed order (deterministic).
#define TZ_KS_TZ 10
__device__ inline double tz_sel4(int blk, double v0, double v1, double v2, double v3) {
  const double a = (blk & 1) ? v1 : v0, b = (blk & 1) ? v3 : v2;
  return (blk & 2) ? b : a;
}
struct TzGStage { double v[TZ_KS_TZ]; double w; int m; };

template <int N>
__device__ inline double tz_row_ror(double v) {                          // value of lane ((lane & 15) - N) mod 16 of the same row of 16
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  constexpr int ctrl = 0x120 + N;
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, ctrl, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), ctrl, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <int R0, int R1>
__device__ inline void tz_gram_rows(const IpmParams& p, double* Hq, const double* Pq, const double* wv, const int* sm, const int h, unsigned long long* pacc) {
  unsigned long long tq0 = pacc ? __builtin_amdgcn_s_memtime() : 0;
  const int lane = tz_tid() & 63;
  const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
  const int Tz = p.Tz, Kc = p.Kc, S = (Kc + 3) >> 2;
  const unsigned rowbytes = (unsigned)(Tz + 1) * 128u;
  const char* gp = (const char*)p.Gp + (unsigned)(4 * k + ij) * 8u;
  double acc[R1 - R0][R1];
#pragma unroll
  for (int a = 0; a < R1 - R0; ++a)
#pragma unroll
    for (int J = 0; J < R1; ++J) acc[a][J] = 0.0;
  auto load = [&](int s, TzGStage& st) {
    int kc = 4 * s + blk; kc = (kc < Kc) ? kc : Kc;                      // patch row Kc is all zero, wv[4 Kc + k] = 0
    st.m = sm[s < S ? s : S];                                            // LDS copy of smask, smask[S] = 0 (made scalar at its use)
    st.w = wv[4 * kc + k];
    const char* prow = gp + (size_t)kc * rowbytes;
#pragma unroll
    for (int J = 0; J < R1; ++J) st.v[J] = tz_ld_pinned((const double*)(prow + (unsigned)(J < Tz ? J : Tz) * 128u));   // tile Tz is zero
  };
  auto mma = [&](const TzGStage& st) {
    const int m = __builtin_amdgcn_readfirstlane(st.m);
#pragma unroll
    for (int I = R0; I < R1; ++I) {
      if ((m >> I) & 1) {
        const double a = st.v[I] * st.w;
#pragma unroll
        for (int J = 0; J <= I; ++J)
          if ((m >> J) & 1) acc[I - R0][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, st.v[J], acc[I - R0][J], 0, 0, 0);
      }
    }
  };
  TzGStage s0, s1;
  load(h, s0);
  for (int s = h; s < S; s += 4) {
    load(s + 2, s1); mma(s0);
    load(s + 4, s0); mma(s1);
  }
  if (pacc) { unsigned long long t1 = __builtin_amdgcn_s_memtime(); pacc[PH_GRAM_LOOP] += t1 - tq0; tq0 = t1; }
  // fold the four patch rows (blk) of every tile with two row rotations (every lane of a row of 16 then holds the tile sum),
  // then store quad by quad: lane (i, blk, j) keeps tile J = 4q + blk, so each quad is ONE full-wave, conflict-free LDS
  // access: H(4I + i, 4(4q + blk) + j).  Tiles right of the diagonal inside the diagonal quad are never read.
  const int i = lane >> 4, j = lane & 3;
#pragma unroll
  for (int a = 0; a < R1 - R0; ++a)
#pragma unroll
    for (int J = 0; J < R1; ++J)
      if (J <= a + R0) { double v = acc[a][J]; v += tz_row_ror<4>(v); v += tz_row_ror<8>(v); acc[a][J] = v; }
  for (int hh = 0; hh < 2; ++hh) {
    if (h == hh) {
#pragma unroll
      for (int I = R0; I < R1; ++I) {
        if (I >= Tz) continue;
#pragma unroll
        for (int q = 0; q <= (I >> 2); ++q) {
          const double v0 = acc[I - R0][4 * q];
          const double v1 = (4 * q + 1 <= I) ? acc[I - R0][(4 * q + 1 <= I) ? 4 * q + 1 : 0] : 0.0;
          const double v2 = (4 * q + 2 <= I) ? acc[I - R0][(4 * q + 2 <= I) ? 4 * q + 2 : 0] : 0.0;
          const double v3 = (4 * q + 3 <= I) ? acc[I - R0][(4 * q + 3 <= I) ? 4 * q + 3 : 0] : 0.0;
          const int idx = (tz_qprefix(I) + q) * TZ_QSTR + TZ_QROW * i + 4 * blk + j;
          Hq[idx] = tz_sel4(blk, v0, v1, v2, v3) + ((hh == 0) ? Pq[idx] : Hq[idx]);
        }
      }
    }
    if (pacc) { unsigned long long t1 = __builtin_amdgcn_s_memtime(); pacc[hh == 0 ? PH_GRAM_RED : PH_GRAM_RMW] += t1 - tq0; tq0 = t1; }
    if (hh == 0) __syncthreads();
    if (pacc && hh == 0) { unsigned long long t1 = __builtin_amdgcn_s_memtime(); pacc[PH_GRAM_BAR] += t1 - tq0; tq0 = t1; }
  }
}

__device__ inline void tz_form_H_ksplit(const IpmParams& p, double* Hq, const double* Pq, const double* wv, const int* sm, unsigned long long* pacc) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = wave >> 1, h = wave & 1;
  if (p.Tz > 7) { if (g == 0) tz_gram_rows<0, 7>(p, Hq, Pq, wv, sm, h, pacc); else tz_gram_rows<7, 10>(p, Hq, Pq, wv, sm, h, pacc); }
  else          { if (g == 0) tz_gram_rows<0, 5>(p, Hq, Pq, wv, sm, h, pacc); else tz_gram_rows<5, 7>(p, Hq, Pq, wv, sm, h, pacc); }
}

__device__ inline void tz_gram(const IpmParams& p, double* Hq, const double* Pq, const double* wv, const int* kl, unsigned long long* pacc = nullptr) {
  if (p.ksplit) tz_form_H_ksplit(p, Hq, Pq, wv, kl, pacc);     // kl holds the super-step masks in this mode
  else tz_form_H(p, Hq, wv, kl);
}

// sqrt(d) and 1/sqrt(d) from v_rsq_f64 + TZ_RSQ_STEPS coupled Newton steps and one residual correction each (deterministic; no
// f64 divide / sqrt sequences).  One step takes the ~2^-26 seed to ~2^-51, the corrections to the last bit or two.
__device__ inline void tz_sqrt_rsqrt(double d, double& sq, double& rs) {
  double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  const double e = __builtin_fma(-g, g, d);
  g = __builtin_fma(e, h, g);
  double inv = h + h;
  const double e2 = __builtin_fma(-g, inv, 1.0);
  inv = __builtin_fma(e2, inv, inv);
  sq = g; rs = inv;
}

// 1/x from v_rcp_f64 (~2^-26) and two Newton steps: five instructions instead of the ~14 of the IEEE division sequence (the kernel
// is issue-bound); to an ulp, for the positive normal x it is used on.
__device__ inline double tz_recip(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  return y;
}

template <int JJ>
__device__ inline double tz_quad_bcast(double v) {                     // value of lane (lane & ~3) + JJ, via DPP quad_perm
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  constexpr int ctrl = JJ * 0x55;
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, ctrl, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), ctrl, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Factor the diagonal tile (pp, pp) (every calling thread redundantly, from LDS) and solve the panel rows of tiles (I, pp),
// I > pp, with the calling threads tid = 0 .. nthr-1.  Thread tid == 0 stores dinv[pp] = inverse of the diagonal factor.
// The diagonal tile itself is not written back: nothing reads it again.
__device__ inline void tz_factor_col(int Tz, double* Hq, double* dinv, int* flag, int pp, int tid, int nthr, unsigned long long* pacc = nullptr) {
  unsigned long long tq0 = pacc ? __builtin_amdgcn_s_memtime() : 0;
  const int dbase = (tz_qprefix(pp) + (pp >> 2)) * TZ_QSTR + 4 * (pp & 3);
  const double a00 = Hq[dbase], a10 = Hq[dbase + TZ_QROW], a11 = Hq[dbase + TZ_QROW + 1];
  const double a20 = Hq[dbase + 2 * TZ_QROW], a21 = Hq[dbase + 2 * TZ_QROW + 1], a22 = Hq[dbase + 2 * TZ_QROW + 2];
  const double a30 = Hq[dbase + 3 * TZ_QROW], a31 = Hq[dbase + 3 * TZ_QROW + 1], a32 = Hq[dbase + 3 * TZ_QROW + 2], a33 = Hq[dbase + 3 * TZ_QROW + 3];
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
  {
    const double m10 = -l10 * i00 * i11;
    const double m21 = -l21 * i11 * i22;
    const double m32 = -l32 * i22 * i33;
    const double m20 = -(l20 * i00 + l21 * m10) * i22;
    const double m31 = -(l31 * i11 + l32 * m21) * i33;
    const double m30 = -(l30 * i00 + l31 * m10 + l32 * m20) * i33;
    if (tid == 0 && !ok) *flag = 1;
    if (tid == 0) {                       // dinv[pp] = M (lower triangular inverse of the diagonal factor); the six zeros above the
      double* m = dinv + pp * 16;         // diagonal are written once per launch by the kernel and never touched again
      m[0] = i00;
      m[4] = m10; m[5] = i11;
      m[8] = m20; m[9] = m21; m[10] = i22;
      m[12] = m30; m[13] = m31; m[14] = m32; m[15] = i33;
    }
  }
  if (pacc) { unsigned long long tq1 = __builtin_amdgcn_s_memtime(); pacc[PH_CH_DIAG] += tq1 - tq0; tq0 = tq1; }
  for (int t = tid; t < 4 * (Tz - pp - 1); t += nthr) {      // x L_pp' = a
    const int I = pp + 1 + (t >> 2), i = t & 3;
    const int base = (tz_qprefix(I) + (pp >> 2)) * TZ_QSTR + TZ_QROW * i + 4 * (pp & 3);
    const double b0 = Hq[base], b1 = Hq[base + 1], b2 = Hq[base + 2], b3 = Hq[base + 3];
    const double x0 = b0 * i00;
    const double x1 = (b1 - x0 * l10) * i11;
    const double x2 = (b2 - x0 * l20 - x1 * l21) * i22;
    const double x3 = (b3 - x0 * l30 - x1 * l31 - x2 * l32) * i33;
    Hq[base] = x0; Hq[base + 1] = x1; Hq[base + 2] = x2; Hq[base + 3] = x3;
  }
  if (pacc) pacc[PH_CH_PANEL] += __builtin_amdgcn_s_memtime() - tq0;
}

// In-place tile-4 LEFT-looking Cholesky.  For tile column pp the tiles (I, pp), I >= pp, are brought up to date with all
// finished columns k < pp in registers (four tile rows per MFMA = the four blocks; two LDS reads per MFMA and no
// read-modify-write of H), written once, then the diagonal tile is factored and the panel solved by all threads.
// Groups of four tile rows are dealt round-robin to the four waves.  Two workgroup barriers per column.
__device__ inline bool tz_cholesky(const IpmParams& p, double* Hq, double* dinv, int* flag, unsigned long long* pacc = nullptr) {
  const int Tz = p.Tz;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;       // operand layout (k', blk, i|j); D layout (i = k, blk, j = ij)
  for (int pp = 0; pp < Tz; ++pp) {
    unsigned long long tc0 = pacc ? __builtin_amdgcn_s_memtime() : 0;
    if (pp > 0) {
      const int pq = pp >> 2, po = 4 * (pp & 3);
      const double* pb = Hq + tz_qprefix(pp) * TZ_QSTR + TZ_QROW * ij + k;         // L(4pp + j, 4k2 + k'), j = ij : + off(k2)
      for (int g = wave; pp + 4 * g < Tz; g += 2 * TZ_NWAVES) {            // two row groups at a time: four independent MFMA chains
        const int Ia = pp + 4 * g + blk, Ib = Ia + 4 * TZ_NWAVES;
        const bool va = Ia < Tz, vb = Ib < Tz;
        const int qa = tz_qprefix(va ? Ia : pp), qb = tz_qprefix(vb ? Ib : pp);
        const double* pa = Hq + qa * TZ_QSTR + TZ_QROW * ij + k;             // L(4Ia + i, 4k2 + k'), i = ij : + off(k2)
        const double* pa2 = Hq + qb * TZ_QSTR + TZ_QROW * ij + k;
        double* pc = Hq + (qa + pq) * TZ_QSTR + TZ_QROW * k + po + ij;       // H(4Ia + i', 4pp + j'), i' = k, j' = ij
        double* pc2 = Hq + (qb + pq) * TZ_QSTR + TZ_QROW * k + po + ij;
        double a0 = *pc, a1 = 0.0, c0 = vb ? *pc2 : 0.0, c1 = 0.0;
        int off = 0, k2 = 0;
        for (; k2 + 1 < pp; k2 += 2) {
          const int off1 = off + 4 + (((k2 & 3) == 3) ? TZ_QSTR - 16 : 0);
          const double nb0 = -pb[off], nb1 = -pb[off1];
          const double x0 = pa[off], x1 = pa[off1], y0 = pa2[off], y1 = pa2[off1];
          a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, nb0, a0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(y0, nb0, c0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, nb1, a1, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, nb1, c1, 0, 0, 0);
          off = off1 + 4 + ((((k2 + 1) & 3) == 3) ? TZ_QSTR - 16 : 0);
        }
        if (k2 < pp) {
          const double nb0 = -pb[off];
          a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa[off], nb0, a0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa2[off], nb0, c0, 0, 0, 0);
        }
        if (va) *pc = a0 + a1;
        if (vb) *pc2 = c0 + c1;
      }
      if (pacc) { unsigned long long t1 = __builtin_amdgcn_s_memtime(); pacc[PH_CH_UPD] += t1 - tc0; tc0 = t1; }
      __syncthreads();
      if (pacc) { unsigned long long t1 = __builtin_amdgcn_s_memtime(); pacc[PH_CH_BAR] += t1 - tc0; tc0 = t1; }
    }
    tz_factor_col(Tz, Hq, dinv, flag, pp, threadIdx.x, TZ_THREADS, pacc);
    if (pacc) tc0 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (pacc) pacc[PH_CH_BAR] += __builtin_amdgcn_s_memtime() - tc0;
  }
  return *flag == 0;
}

// The same factorisation by ONE wave (wave-level ordering of its own LDS traffic instead of workgroup barriers), for matrices
// of at most 16 tile columns: the other three waves are free to run an independent G' product meanwhile (tz_ipm_kernel).
__device__ inline void tz_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Bounded wait of one wave for a counter in LDS that other waves of the workgroup advance (release stores / adds): true when
// *f >= target was seen.  The bound is a safety net (the producers never wait for this wave): ~1e6 polls.
__device__ inline bool tz_spin_until(const int* f, int target) {
  for (int guard = 0; guard < (1 << 20); ++guard) {
    const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (v >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// prog (LDS, may be null): number of finished tile columns, published after each one -- column pp of L and inv(L_pp,pp) are final
// then, which is all the forward substitution trailing on another wave (tz_fwd_trailing) needs for its step pp.
__device__ inline void tz_cholesky_wave(const IpmParams& p, double* Hq, double* dinv, int* flag, int* prog = nullptr) {
  const int Tz = p.Tz;
  const int lane = tz_tid() & 63;
  const int k = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
  for (int pp = 0; pp < Tz; ++pp) {
    if (pp > 0) {
      const int pq = pp >> 2, po = 4 * (pp & 3);
      const double* pb = Hq + tz_qprefix(pp) * TZ_QSTR + TZ_QROW * ij + k;
      for (int g = 0; pp + 4 * g < Tz; g += 2) {
        const int Ia = pp + 4 * g + blk, Ib = Ia + 4;
        const bool va = Ia < Tz, vb = Ib < Tz;
        const int qa = tz_qprefix(va ? Ia : pp), qb = tz_qprefix(vb ? Ib : pp);
        const double* pa = Hq + qa * TZ_QSTR + TZ_QROW * ij + k;
        const double* pa2 = Hq + qb * TZ_QSTR + TZ_QROW * ij + k;
        double* pc = Hq + (qa + pq) * TZ_QSTR + TZ_QROW * k + po + ij;
        double* pc2 = Hq + (qb + pq) * TZ_QSTR + TZ_QROW * k + po + ij;
        double a0 = *pc, a1 = 0.0, c0 = vb ? *pc2 : 0.0, c1 = 0.0;
        int off = 0, k2 = 0;
        for (; k2 + 1 < pp; k2 += 2) {
          const int off1 = off + 4 + (((k2 & 3) == 3) ? TZ_QSTR - 16 : 0);
          const double nb0 = -pb[off], nb1 = -pb[off1];
          const double x0 = pa[off], x1 = pa[off1], y0 = pa2[off], y1 = pa2[off1];
          a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, nb0, a0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(y0, nb0, c0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, nb1, a1, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, nb1, c1, 0, 0, 0);
          off = off1 + 4 + ((((k2 + 1) & 3) == 3) ? TZ_QSTR - 16 : 0);
        }
        if (k2 < pp) {
          const double nb0 = -pb[off];
          a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa[off], nb0, a0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa2[off], nb0, c0, 0, 0, 0);
        }
        if (va) *pc = a0 + a1;
        if (vb) *pc2 = c0 + c1;
      }
      tz_wave_sync();
    }
    tz_factor_col(Tz, Hq, dinv, flag, pp, lane, 64);
    tz_wave_sync();
    if (prog && lane == 0) __hip_atomic_store(prog, pp + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// Solve (L L') out = rhs.  Thread t owns row t (nzp <= 256) in a register; wave w owns the 64-row block w.
// Forward: the blocks are finished one after the other.  Inside a block the 16 tile steps run wave-synchronously (owner
// quad finishes its 4 unknowns with DPP broadcasts and publishes them in LDS, the other lanes of the SAME wave pick them up:
// in-wave LDS ordering, no workgroup barrier); between blocks one barrier and one bulk update of the rows of later blocks.
// Backward: the mirror image.  ybuf: nzp doubles of LDS.
__device__ inline void tz_chol_solve(const IpmParams& p, const double* Hq, const double* dinv,
                                     const double* rhs, double* ybuf, double* out) {
  const int Tz = p.Tz, nzp = p.nzp;
  const int t = tz_tid(), jq = t & 3, tq = t >> 2, wave = t >> 6;
  const int nblk = (Tz + 15) >> 4;
  double rv = (t < nzp) ? rhs[t] : 0.0;
  const int rowbase = tz_qprefix(tq) * TZ_QSTR + TZ_QROW * jq;                    // + (I>>2)*64 + 4(I&3): L(t, 4I + .)
  for (int blkI = 0; blkI < nblk; ++blkI) {                             // ---- forward: L y = rhs
    const int I0 = 16 * blkI, I1 = min(Tz, I0 + 16);
    if (wave == blkI) {
      for (int I = I0; I < I1; ++I) {
        if (tq == I) {
          const double a0 = tz_quad_bcast<0>(rv), a1 = tz_quad_bcast<1>(rv), a2 = tz_quad_bcast<2>(rv), a3 = tz_quad_bcast<3>(rv);
          const double* m = dinv + I * 16 + 4 * jq;                      // row jq of M (lower)
          rv = m[0] * a0 + m[1] * a1 + m[2] * a2 + m[3] * a3;
          ybuf[t] = rv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tq > I && t < nzp) {
          const int base = rowbase + (I >> 2) * TZ_QSTR + 4 * (I & 3);
          const double* y = ybuf + 4 * I;
          rv -= Hq[base] * y[0] + Hq[base + 1] * y[1] + Hq[base + 2] * y[2] + Hq[base + 3] * y[3];
        }
      }
    }
    if (blkI + 1 < nblk) {
      __syncthreads();
      if (wave > blkI && t < nzp) {                                      // bulk: rows of later blocks take this block's y
        double acc = 0.0;
        for (int I = I0; I < I1; ++I) {
          const int base = rowbase + (I >> 2) * TZ_QSTR + 4 * (I & 3);
          const double* y = ybuf + 4 * I;
          acc += Hq[base] * y[0] + Hq[base + 1] * y[1] + Hq[base + 2] * y[2] + Hq[base + 3] * y[3];
        }
        rv -= acc;
      }
    }
  }
  const int colbase = (t >> 4) * TZ_QSTR + 4 * ((t >> 2) & 3) + (t & 3);     // + qprefix(I)*64 + 16k: L(4I + k, t)
  for (int blkI = nblk - 1; blkI >= 0; --blkI) {                        // ---- backward: L' x = y
    const int I0 = 16 * blkI, I1 = min(Tz, I0 + 16);
    if (wave == blkI) {
      for (int I = I1 - 1; I >= I0; --I) {
        if (tq == I) {
          const double y0 = tz_quad_bcast<0>(rv), y1 = tz_quad_bcast<1>(rv), y2 = tz_quad_bcast<2>(rv), y3 = tz_quad_bcast<3>(rv);
          const double* m = dinv + I * 16 + jq;                          // column jq of M
          rv = m[0] * y0 + m[4] * y1 + m[8] * y2 + m[12] * y3;
          ybuf[t] = rv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tq < I && tq >= I0) {
          const int base = tz_qprefix(I) * TZ_QSTR + colbase;
          const double* x = ybuf + 4 * I;
          rv -= Hq[base] * x[0] + Hq[base + TZ_QROW] * x[1] + Hq[base + 2 * TZ_QROW] * x[2] + Hq[base + 3 * TZ_QROW] * x[3];
        }
      }
    }
    if (blkI > 0) {
      __syncthreads();
      if (wave < blkI) {                                                 // bulk: rows of earlier blocks take this block's x
        double acc = 0.0;
        for (int I = I0; I < I1; ++I) {
          const int base = tz_qprefix(I) * TZ_QSTR + colbase;
          const double* x = ybuf + 4 * I;
          acc += Hq[base] * x[0] + Hq[base + TZ_QROW] * x[1] + Hq[base + 2 * TZ_QROW] * x[2] + Hq[base + 3 * TZ_QROW] * x[3];
        }
        rv -= acc;
      }
    }
  }
  if (t < nzp) out[t] = rv;
}

// (L L') out = rhs for matrices of at most 16 tile columns (nzp <= 64): wave 0 alone, lane = row, the value of every unknown
// travels by v_readlane instead of an LDS publish + wave sync + read -- per tile step the dependent chain is four DPP quad
// broadcasts, the 4x4 inverse-diagonal product, four readlanes and one fused update; the L / M entries of the next step are
// independent LDS reads.  Called by all threads (waves 1-3 fall through); the caller's barrier publishes out.
// bwd_only: rhs already holds y = inv(L) r (left there by tz_fwd_trailing).
__device__ inline void tz_chol_solve_wave(const IpmParams& p, const double* Hq, const double* dinv, const double* rhs, double* out, bool bwd_only = false) {
  const int tt = tz_tid();
  if (tt >= 64) return;
  __builtin_amdgcn_s_setprio(TZ_PRIO_SOLVE);
  const int Tz = p.Tz, nzp = p.nzp;
  const int t = tt, jq = t & 3, tq = t >> 2;
  double rv = (t < nzp) ? rhs[t] : 0.0;
  const int rowbase = tz_qprefix(tq < Tz ? tq : 0) * TZ_QSTR + TZ_QROW * jq;       // + (I>>2)*QSTR + 4(I&3): L(t, 4I + .)
  // operands of tile step I are independent of the running solution: those of the next step are fetched while this one computes
  auto fwd_load = [&](int I, double (&mm)[4], double (&ll)[4]) {
    const int Ic = I < Tz ? I : Tz - 1;
    const double* m = dinv + Ic * 16 + 4 * jq;                         // row jq of M_I = inv(L_II)
    const int base = rowbase + (Ic >> 2) * TZ_QSTR + 4 * (Ic & 3);
    const bool below = tq > Ic && tq < Tz;
#pragma unroll
    for (int k = 0; k < 4; ++k) { mm[k] = m[k]; ll[k] = below ? Hq[base + k] : 0.0; }
  };
  double mA[4], lA[4], mB[4], lB[4];
  auto fwd_step = [&](int I, const double (&mm)[4], const double (&ll)[4]) {
    const double a0 = tz_quad_bcast<0>(rv), a1 = tz_quad_bcast<1>(rv), a2 = tz_quad_bcast<2>(rv), a3 = tz_quad_bcast<3>(rv);
    const double yc = (mm[0] * a0 + mm[1] * a1) + (mm[2] * a2 + mm[3] * a3);       // y of this quad if it is the owner (tq == I)
    const double y0 = tz_readlane(yc, 4 * I), y1 = tz_readlane(yc, 4 * I + 1), y2 = tz_readlane(yc, 4 * I + 2), y3 = tz_readlane(yc, 4 * I + 3);
    rv = (tq == I) ? yc : rv - ((ll[0] * y0 + ll[1] * y1) + (ll[2] * y2 + ll[3] * y3));
  };
  if (!bwd_only) {
  fwd_load(0, mA, lA);
  for (int I = 0; I < Tz; I += 2) {                                    // ---- forward: L y = rhs
    fwd_load(I + 1, mB, lB);
    fwd_step(I, mA, lA);
    fwd_load(I + 2, mA, lA);
    if (I + 1 < Tz) fwd_step(I + 1, mB, lB);
  }
  }
  const int colbase = (t >> 4) * TZ_QSTR + 4 * ((t >> 2) & 3) + (t & 3);        // + qprefix(I)*QSTR + QROW k: L(4I + k, t)
  auto bwd_load = [&](int I, double (&mm)[4], double (&ll)[4]) {
    const int Ic = I >= 0 ? I : 0;
    const double* m = dinv + Ic * 16 + jq;                             // column jq of M_I
    const int base = tz_qprefix(Ic) * TZ_QSTR + colbase;
    const bool above = tq < Ic;
#pragma unroll
    for (int k = 0; k < 4; ++k) { mm[k] = m[4 * k]; ll[k] = above ? Hq[base + k * TZ_QROW] : 0.0; }
  };
  bwd_load(Tz - 1, mA, lA);
  for (int I = Tz - 1; I >= 0; I -= 2) {                               // ---- backward: L' x = y
    bwd_load(I - 1, mB, lB);
    fwd_step(I, mA, lA);                                               // same arithmetic with the transposed operands
    bwd_load(I - 2, mA, lA);
    if (I - 1 >= 0) fwd_step(I - 1, mB, lB);
  }
  if (t < nzp) out[t] = rv;
  __builtin_amdgcn_s_setprio(0);
}

// Forward substitution L y = r by ONE wave (any: lane = row) while another wave is still factoring: step I starts when the
// factorisation has published column I (prog > I).  rv: this lane's entry of r; y goes to yout (LDS, nzp doubles).  Returns
// false if the wait ran into its bound (never seen; the caller then fails the solve).
__device__ inline bool tz_fwd_trailing(const IpmParams& p, const double* Hq, const double* dinv, double rv, const int* prog, double* yout) {
  const int Tz = p.Tz, nzp = p.nzp;
  const int t = tz_tid() & 63, jq = t & 3, tq = t >> 2;
  const int rowbase = tz_qprefix(tq < Tz ? tq : 0) * TZ_QSTR + TZ_QROW * jq;
  bool ok = true;
  for (int I = 0; I < Tz; ++I) {
    if (!tz_spin_until(prog, I + 1)) { ok = false; break; }
    const double* m = dinv + I * 16 + 4 * jq;                          // row jq of M_I = inv(L_II)
    const int base = rowbase + (I >> 2) * TZ_QSTR + 4 * (I & 3);
    const bool below = tq > I && tq < Tz;
    double mm[4], ll[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mm[k] = m[k]; ll[k] = below ? Hq[base + k] : 0.0; }
    const double a0 = tz_quad_bcast<0>(rv), a1 = tz_quad_bcast<1>(rv), a2 = tz_quad_bcast<2>(rv), a3 = tz_quad_bcast<3>(rv);
    const double yc = (mm[0] * a0 + mm[1] * a1) + (mm[2] * a2 + mm[3] * a3);
    const double y0 = tz_readlane(yc, 4 * I), y1 = tz_readlane(yc, 4 * I + 1), y2 = tz_readlane(yc, 4 * I + 2), y3 = tz_readlane(yc, 4 * I + 3);
    rv = (tq == I) ? yc : rv - ((ll[0] * y0 + ll[1] * y1) + (ll[2] * y2 + ll[3] * y3));
  }
  if (t < nzp) yout[t] = rv;
  return ok;
}

#include "tz_tt.hip.h"
// unit size of the blocked Gram by register budget (MINW = workgroups per CU the variant is compiled for: 2 -> 256 registers,
// 1 -> 512) and super-steps in flight; the host plans its units with the same size (tzddpc_hip.hip)
#define TZ_TT_NST 4
#define TZ_TT_GU(minw) ((minw) >= 2 ? 6 : 8)
// triangular solves of the tile-triangle class: by 16 x 16 diagonal blocks with explicit block inverses (default) or tile by tile
#define TZ_TT_AFTER_CHOL(p, H, dinv) do { tz_tt_block_inverse(p, H, dinv); __syncthreads(); } while (0)
#define TZ_TT_SOLVE(p, H, dinv, rhs, ybuf, out) tz_chol_solve_blk(p, H, rhs, ybuf, out)

// LDS footprint in doubles (host mirrors this in tzddpc_hip.hip).  hsize: doubles of the factor storage -- nquads * TZ_QSTR in the
// quad layout (nz <= 64), ntile * TS in the tile-triangle layout; the latter keeps 16 more doubles behind dinv for the factor of
// the diagonal tile being eliminated (tz_cholesky_tt).
__host__ __device__ inline size_t tz_ipm_lds_doubles(size_t hsize, int tt, int Tz, int nzp, int mip, int nklist, int ntheta, int ksplit, int ntube, int nell, int park = 0) {
  return (park ? 2 * (size_t)mip : 0) + (ksplit ? hsize : 0) + hsize + (size_t)Tz * 16 + (tt ? 16 : 0) + 14 * (size_t)nzp + (size_t)(mip + 4) + 32 + 2 + (size_t)((nklist + 1) / 2) + (size_t)ntheta + 4 * TZ_NMAX + (size_t)ntube + (size_t)nell;
}

typedef __attribute__((address_space(4))) const IpmParams* TzKargPtr;

// MINW = workgroups the kernel is compiled to fit on one CU's register file (4: 128 VGPRs; 2: 256; 1: 512).  Problems whose
// LDS footprint allows fewer than four workgroups per CU anyway get the variant with the larger register budget (no spills).
template <int MAXR, int NCG, int MINW = TZ_MINWAVES>
__global__ __launch_bounds__(TZ_THREADS, MINW) void tz_ipm_kernel(IpmParams p) {
#if TZ_PROFILE
  const bool PROF = p.prof != nullptr && blockIdx.x == 0;
#else
  constexpr bool PROF = false;            // the diagnostic build (libtzddpc_hip_prof.so, -DTZ_PROFILE=1) carries the per-phase clocks
#endif
  unsigned long long tprev = 0, tstart = 0;
  unsigned long long acc_ph[PH_COUNT] = {};
#define TZ_STAMP(ph) do { if (PROF) { unsigned long long _t = __builtin_amdgcn_s_memtime(); acc_ph[ph] += _t - tprev; tprev = _t; } } while (0)
  if (PROF) { tprev = __builtin_amdgcn_s_memtime(); tstart = tprev; }
  extern __shared__ double lds[];
  const int b = blockIdx.x;
  int t = threadIdx.x;                     // re-laundered at phase boundaries (TZ_FRESH_T, see tz_tid)
#define TZ_FRESH_T() asm volatile("" : "+v"(t))
  // columns: nz <= nzp <= 256 = TZ_THREADS (limit of tz_problem_create), so "every column" is one predicated statement, not a loop
#define TZ_COLS(c, lim) if (const int c = t; c < (lim))
  const bool wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
  const bool wave1 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 1;
  const int nz = p.nz, mi = p.mi, nzp = p.nzp, mip = p.mip;
  const FuseParams& F0 = p.F;
  const bool fused = F0.on != 0;          // closed-loop step in one launch: tube + parameter maps before, recovery + plant after
  if (!fused && p.prestatus[b] != 0) {
    if (t == 0) { p.status[b] = 3; p.iters[b] = 0; if (p.status_copy) p.status_copy[b] = 3; }
    TZ_COLS(c, nz) p.x[(size_t)b * nz + c] = 0.0;
    for (int r = t; r < mi; r += TZ_THREADS) { p.s[(size_t)b * mi + r] = 1.0; p.lam[(size_t)b * mi + r] = 0.0; }
    return;
  }
  constexpr bool TT = (NCG >= 2);          // more than 64 variables: tile-triangle layout, blocked Gram, two-phase Cholesky (tz_tt.hip.h)
  const size_t hsize = TT ? (size_t)p.ntile * p.TS : (size_t)p.nquads * TZ_QSTR;
  double* Hq = lds;
  double* dinv = Hq + hsize;
  double* dfac = dinv + p.Tz * 16;         // TT only
  double* xv = dfac + (TT ? 16 : 0);
  double* dxv = xv + nzp;
  double* rdv = dxv + nzp;
  double* r1v = rdv + nzp;
  double* qv = r1v + nzp;
  double* tmpz = qv + nzp;
  double* part = tmpz + nzp;              // 4 * nzp
  double* part2 = part + 4 * nzp;         // 4 * nzp
  double* vin = part2 + 4 * nzp;           // mip + 4 : staging of one row vector (w for the Gram, inputs of G' products)
  double* red = vin + mip + 4;            // 32: two exchange buffers of the block reductions (+ the stopping-test scales in [12], [13])
  int* flag = (int*)(red + 32);
  int* kl = (int*)(red + 34);
  double* thl = red + 34 + (p.nklist + 1) / 2;      // theta of this trajectory (fused step only)
  double* stl = thl + p.F.ntheta;                   // fused: closed-loop state [x | xbar | e] (3 n doubles), disturbance of the step at 3 TZ_NMAX
  double* pl = stl + 4 * TZ_NMAX;                   // partial sums of the lane-ELL products (nell doubles)
  double* tbl = pl + p.nell;                  // fused: C_K powers and the tube resolvent (copied once per launch)
  double* Pq = tbl + p.ntube;                       // ksplit: P + reg I in the quad layout of Hq (lower tiles)
  // 128-register variant: h and G x of the rows live in LDS (each thread touches only its own slots: no barrier) -- two
  // register pairs fewer across the whole solve
  constexpr bool PARK = (MINW == 4);
  double* hL = Pq + (p.ksplit ? hsize : 0);
  double* gL = hL + mip;

  // rows owned by this thread
  double s_[MAXR], l_[MAXR], h_[MAXR], gx_[MAXR];        // live across iterations (s, lambda also across steps); h, gx unused when parked
#define TZ_H(k, r) (PARK ? hL[r] : h_[k])
#define TZ_GX(k, r) (PARK ? gL[r] : gx_[k])
#define TZ_SET_H(k, r, v) do { if (PARK) hL[r] = (v); else h_[k] = (v); } while (0)
#define TZ_ADD_GX(k, r, v) do { if (PARK) gL[r] += (v); else gx_[k] += (v); } while (0)
#define TZ_ROWS(k, r) _Pragma("unroll") for (int k = 0; k < MAXR; ++k) if (const int r = t + TZ_THREADS * k; r < mi)
  int rseg_[MAXR];                                   // lanes of the G x product that carry this thread's rows
#pragma unroll
  for (int k = 0; k < MAXR; ++k) { const int r = t + TZ_THREADS * k; rseg_[k] = (r < mi) ? p.eg.seg[r] : 0; }
  // lanes of the G'v product that carry column t; for nz <= 64 wave 1 holds a second copy (column t - 64): it assembles the
  // predictor's right-hand side while wave 0 factors
  const int ccol = (nzp <= 64 && t >= 64) ? (t < 128 ? t - 64 : nz) : t;
  const int cseg = (ccol < nz) ? p.et.seg[ccol] : 0;

  // ---- once per launch: constants of the problem into LDS, closed-loop state of the trajectory -------------------------
  if (TT) { (void)kl; (void)Pq; }
  else if (p.ksplit) {
    for (int i = t; i <= ((p.Kc + 3) >> 2); i += TZ_THREADS) kl[i] = p.smask[i];
    for (int e = t; e < p.nquads * 64; e += TZ_THREADS) {                // every entry of every quad (padding tiles: 0)
      const int qd = e >> 6;
      int I = 0;
      while (tz_qprefix(I + 1) <= qd) ++I;
      const int r = 4 * I + ((e >> 4) & 3), c = 4 * (4 * (qd - tz_qprefix(I)) + ((e >> 2) & 3)) + (e & 3);
      double v = (c < nzp) ? p.P[(size_t)r * nzp + c] : 0.0;
      if (r == c) v = (r < nz) ? v + p.reg : 1.0;
      Pq[qd * TZ_QSTR + TZ_QROW * ((e >> 4) & 3) + (e & 15)] = v;
    }
  }
  else for (int i = t; i < p.nklist; i += TZ_THREADS) kl[i] = p.klist[i];
  if (fused) {
    const int n = F0.fin.n, m = F0.fin.m, nv = F0.fin.N * m;
    const int n1 = (F0.tube.pmax + 1) * n * n, n2 = n1 + (F0.tube.pmax > 0 ? F0.tube.pmax : 1) * (n + m) * n;
    for (int i = t; i < n2; i += TZ_THREADS) tbl[i] = (i < n1) ? F0.tube.CKpow[i] : F0.tube.T[i - n1];
    // constants of the recovery / plant update: [A | B | K | r1 | R2 | Phi rows of xbar[1] | Gam rows of xbar[1] | Dz(v)]
    double* ec = tbl + n2;
    for (int i = t; i < n * n; i += TZ_THREADS) { ec[i] = F0.plant.A[i]; ec[2 * n * m + n * n + n + i] = F0.fin.R2[i]; ec[2 * n * m + 2 * n * n + n + i] = F0.fin.Phi[(size_t)n * n + i]; }
    for (int i = t; i < n * m; i += TZ_THREADS) { ec[n * n + i] = F0.plant.Bm[i]; ec[n * n + n * m + i] = F0.plant.K[i]; }
    for (int i = t; i < n; i += TZ_THREADS) ec[n * n + 2 * n * m + i] = F0.fin.r1[i];
    for (int i = t; i < n * nv; i += TZ_THREADS) ec[3 * n * n + 2 * n * m + n + i] = F0.fin.Gam[(size_t)n * nv + i];
    for (int i = t; i < nv; i += TZ_THREADS) ec[3 * n * n + 2 * n * m + n + n * nv + i] = F0.fin.Dz[i];
    int* pwl = (int*)(ec + 3 * n * n + 2 * n * m + n + n * nv + nv);                // power[k] of the N steps (tube)
    for (int i = t; i < F0.fin.N; i += TZ_THREADS) pwl[i] = F0.tube.power[i];
  }
  if (fused && t < F0.fin.n) {                    // closed-loop state [x | xbar | e] stays in LDS for all steps of this launch
    stl[t] = F0.plant.x[(size_t)b * F0.fin.n + t];
    stl[F0.fin.n + t] = F0.plant.xbar[(size_t)b * F0.fin.n + t];
    stl[2 * F0.fin.n + t] = F0.plant.e[(size_t)b * F0.fin.n + t];
  }
#pragma unroll
  for (int k = 0; k < MAXR; ++k) { s_[k] = 1.0; l_[k] = 0.0; h_[k] = 0.0; gx_[k] = 0.0; }
  TZ_COLS(c, nzp) xv[c] = 0.0;
  for (int c = t; c < p.Tz * 16; c += TZ_THREADS) dinv[c] = 0.0;
  for (int r = t; r < mip + 4; r += TZ_THREADS) vin[r] = 0.0;          // rows written per use; the pad behind row mi is read by the Gram and stays zero
  __syncthreads();

  const int nsteps = fused ? F0.nsteps : 1;
  if (fused && F0.sticky_fresh != 0 && t == 0 && F0.plant.sticky) F0.plant.sticky[b] = 0;     // this launch owns the first-failure record (read back by the same thread)
  int was_shifted = (p.warm != 0 && (p.shift_policy >= 2 || p.shift_state[b] == -2)) ? p.shift_state[b] : 0;
  int status = 1, it = 0;
  int work_f = 0, work_s = 0;                // uniform: kept in scalar registers
  int rlev_carry = 0;                        // diagonal-shift level the previous solved step of this launch ended with ...
  unsigned brk_hi = 0;                       // ... and (the high word of) TZ_RLEV_GATE x the complementarity at which it first broke down
  int rpar = 0;                              // which exchange buffer the next block reduction uses
  for (int step = 0; step < nsteps; ++step) {     // closed-loop steps of this trajectory (one when the launch is a single solve)
  // start point of this step: 0 cold, 1 the (x, lambda) stored by an earlier launch, 2 the (x, lambda) of the previous step (still
  // in LDS / registers)
  TZ_FRESH_T();
  // The fused-step arguments are needed only before and after the interior point: re-read them from the kernel argument
  // segment here and in the epilogue (opaque pointer: the loads cannot be hoisted out of the step loop, so the ~120 scalar
  // registers they would pin are free during the solve).  The same goes for the other arguments that only the start and the
  // end of a step touch (stored solution, shift maps, warm-start switches): `pk` here, `pe` at the end, `pi` for the tolerances.
  TzKargPtr kp0 = (TzKargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp0));
  const FuseParams& F = ((const IpmParams*)kp0)->F;
  const IpmParams& pk = *(const IpmParams*)kp0;
  int src = 0;
  if (step == 0) src = (pk.warm != 0 && pk.prev_status != nullptr && pk.prev_status[b] == 0) ? 1 : 0;
  else src = (pk.warm_steps != 0 && status == 0) ? 2 : 0;
  bool skip = false;                        // fused step: a parameter row is violated -> status 3, u = K e, nominal state from Phi
  // disturbance of this step: needed only by the plant update at the very end, but a cold line (every step of every trajectory
  // has its own) -- fetched now, parked in LDS at the end of the prologue
  double wpre = 0.0;
  if (fused && t < F.fin.n) wpre = F.plant.w[(size_t)b * F.plant.w_stride + (size_t)step * F.w_step + t];
  // warm-start policy of this step (uniform): the previous solution, optionally moved one step along the horizon (v_k <- v_{k+1} ...:
  // better in a transient, a matter of the problem otherwise -- tz_problem_set_warm_shift)
  // was_shifted: 1 + q = the last start was shifted, after q quiet (<= 1 iteration) shifted steps; 0 = not shifted; -1 = not shifted because
  // the quiet budget ran out ("rest", round 4): the unshifted start is kept for as long as it needs no iteration at all -- a settled loop:
  // no shift, no G x, the cheapest step there is -- and the shifted regime is re-entered the moment a step needs one.  (Until round 4
  // a trajectory that left the shifted regime came back only after a step of >= k iterations: a loop that was not settled yet then
  // took one Newton step in EVERY step from the 17th quiet step on -- 0.51 factorisations per step in steps 30-130 of the headline
  // problem instead of 0.01.)
  bool shifted = false;
  const bool seeded = src != 0 && was_shifted == -2;
  if (seeded) {
    // the stored start (tz_problem_store_start) IS a solution for the start of the loop: taken as it is, and counted as the first
    // quiet step of the shifted regime (a loop that begins with a step of 0 iterations would otherwise never enter it)
    was_shifted = (pk.shift_policy >= 2) ? 1 : 0;
  } else if (src != 0) {
    const int prev_it = (step == 0) ? pk.iters[b] : it;
    const bool quiet_run = was_shifted > 0 && prev_it <= 1;
    const int nquiet = quiet_run ? was_shifted : 0;                       // was_shifted - 1 quiet steps so far, this one included: was_shifted
    const bool back = was_shifted < 0 && prev_it >= 1;
    shifted = pk.shift_policy == 1 || (pk.shift_policy >= 2 && (prev_it >= pk.shift_policy || back || (quiet_run && (pk.shift_quiet == 0 || nquiet <= pk.shift_quiet))));
    was_shifted = shifted ? 1 + nquiet : (((quiet_run && pk.shift_policy >= 2) || (was_shifted < 0 && !back)) ? -1 : 0);
  }
  // A step that starts from the previous step of the same launch (src == 2: x in LDS, lambda in registers) runs the warm start INSIDE the
  // barrier intervals of the prologue: the two halves of the horizon shift beside the two stages of the tube pass, the walk of G x
  // beside the parameter maps (neither depends on theta) -- four workgroup barriers from the plant update of one step to the
  // warm-start reduction of the next instead of seven.  G x of the starting point: inside a launch gx_ still holds it (it followed x
  // through the iterations of the previous step); it is formed afresh every eighth step (rounding) and whenever the point was shifted.
  const bool inlaunch = fused && src == 2;
  const bool walk_early = inlaunch && (shifted || (step & 7) == 0);
  if (fused) {
    if (t == 0) { flag[0] = 0; flag[1] = 0; }
    const int n = F.fin.n, m = F.fin.m, nv = F.fin.N * m;
    const double* Tt = tbl + (F.tube.pmax + 1) * n * n;
    const int* pwl = (const int*)(Tt + (F.tube.pmax > 0 ? F.tube.pmax : 1) * (n + m) * n + 3 * n * n + 2 * n * m + n + n * nv + nv);
    // the factor storage is free until the first Gram: |C_K^l e0| goes there
    if (n == 2 && m == 1) tz_tube_stage1<2, 1>(F.tube, tbl, stl + n, stl + 2 * n, Hq, thl, t, TZ_THREADS);
    else tz_tube_stage1(F.tube, tbl, stl + n, stl + 2 * n, Hq, thl, t, TZ_THREADS);
    if (inlaunch && shifted) {
      TZ_COLS(c, nz) tmpz[c] = xv[pk.sx[c]] * pk.sxs[c];
      TZ_ROWS(k, r) vin[r] = l_[k];
    }
    __syncthreads();
    if (n == 2 && m == 1) tz_tube_stage2<2, 1>(F.tube, tbl, Tt, pwl, stl + 2 * n, Hq, thl, t, TZ_THREADS);
    else tz_tube_stage2(F.tube, tbl, Tt, pwl, stl + 2 * n, Hq, thl, t, TZ_THREADS);
    if (inlaunch && shifted) {
      TZ_COLS(c, nz) xv[c] = tmpz[c];
      TZ_ROWS(k, r) l_[k] = vin[pk.sr[r]] * pk.sls[r];
    }
    __syncthreads();
    TZ_STAMP(PH_TUBE);
    TZ_COLS(c, nzp) { qv[c] = (c < nz) ? csr_row(F.qmap, c, thl) : 0.0; if (src != 2) xv[c] = 0.0; }
    TZ_STAMP(PH_MAPS_Q);
    int bad = 0;
    for (int r = t; r < F.npar; r += TZ_THREADS) {
      const double v = csr_row(F.parmap, r, thl);
      if (!(v >= F.par_lo[r] - 1e-9) || !(v <= F.par_hi[r] + 1e-9)) bad = 1;
    }
    if (bad) flag[1] = 1;
    TZ_ROWS(k, r) TZ_SET_H(k, r, csr_row(F.hmap, r, thl));
    if (walk_early) tz_ell_gemv_walk<TT>(p, xv, pl);         // (pl: last read before the barriers above)
  } else {
    TZ_COLS(c, nzp) qv[c] = (c < nz) ? pk.q[(size_t)b * nz + c] : 0.0;
    TZ_ROWS(k, r) TZ_SET_H(k, r, pk.h[(size_t)b * mi + r]);
    if (t == 0) *flag = 0;
  }
  if (src != 2) { TZ_ROWS(k, r) l_[k] = 1.0; }
  if (fused && t < F.fin.n) stl[3 * TZ_NMAX + t] = wpre;
  __syncthreads();
  if (fused && flag[1] != 0) { skip = true; TZ_ROWS(k, r) { s_[k] = 1.0; l_[k] = 0.0; } TZ_COLS(c, nzp) xv[c] = 0.0; }
  TZ_STAMP(PH_PROLOGUE);

  // exact dual residual rd = P x + q + G'lam into rdv (used at the start and to confirm convergence)
  // staged == true: vin already holds lambda and a barrier has passed since (the warm start stages it; the reduction at the top of
  // the first iteration is the barrier).  Returns max |rd| (every thread).
  auto exact_rd = [&](bool staged) -> double {
    TZ_STAMP(PH_TEST);
    if (!staged) {
      TZ_ROWS(k, r) vin[r] = l_[k];
      __syncthreads();
    }
    TZ_STAMP(PH_RD_A);
    if (wave0) tz_gemvT_partial<NCG, 0, 1>(p.P, p.nP, nzp, xv, part);     // P x by wave 0 (stays in `part` for the objective)
    else tz_ell_gemvT_part<TT>(p, vin, pl);                                   // G'lambda by waves 1-3
    __syncthreads();
    TZ_STAMP(PH_RD_B);
    double e1 = 0.0;
    if (nzp <= 64) {
      // all columns sit in wave 0: its maximum is the workgroup's -- one wave reduction, one LDS hop, one barrier (instead of store,
      // barrier, re-read, four wave reductions, exchange, barrier)
      if (wave0) {
        const double v = (t < nz) ? (tz_ell_colsum(pl, cseg) + qv[t]) + part[t] : 0.0;
        if (t < nzp) rdv[t] = v;
        e1 = tz_wave_reduce<RED_MAX>(fabs(v));
        if (t == 0) red[14] = e1;                      // (rewritten by the next test only: there is always a barrier in between)
      }
      __syncthreads();
      e1 = red[14];
    } else {
      if (t < nzp) rdv[t] = (t < nz) ? (tz_ell_colsum(pl, cseg) + qv[t]) + part[t] : 0.0;
      __syncthreads();
      double e2 = 0, e3 = 0;
      TZ_COLS(c, nz) e1 = fmax(e1, fabs(rdv[c]));
      tz_block_reduce3<RED_MAX, RED_MAX, RED_MAX, 1>(e1, e2, e3, red, rpar);
    }
    TZ_STAMP(PH_RD_C);
    return e1;
  };

  // A solve that does not end in TZ_SOLVED (in practice: the aggressive fraction to the boundary collapsing mu before the
  // residuals on a degenerate problem, ~2e-5 of the pulley steps) is repeated once from a cold start with the textbook 0.99.
  int attempt = 0;
retry_solve:
  const bool retried = attempt != 0;
  bool okf = true;
  bool warm = !skip && src != 0;     // the previous step of this trajectory was solved: start from it
  double scq = 0, sch = 0;
  if (!skip) {
  if (warm) {
    // ---- warm start: previous (x, lambda) of this trajectory, slacks re-derived for the new h and pushed into the cone
    // by at least the amount the old point violates the new rows; a point that is too far outside starts cold instead
    if (src == 1) {
      if (shifted) {
        TZ_COLS(c, nz) xv[c] = pk.x[(size_t)b * nz + pk.sx[c]] * pk.sxs[c];
        TZ_ROWS(k, r) l_[k] = pk.lam[(size_t)b * mi + pk.sr[r]] * pk.sls[r];
      } else {
        TZ_COLS(c, nz) xv[c] = pk.x[(size_t)b * nz + c];
        TZ_ROWS(k, r) l_[k] = pk.lam[(size_t)b * mi + r];
      }
    }
    if (walk_early && !retried) { tz_ell_gemv_sum<MAXR>(pl, rseg_, gx_); if (PARK) { TZ_ROWS(k, r) gL[r] = gx_[k]; } }     // walked in the prologue
    else if (src != 2 || retried) { tz_ell_gemv<MAXR, TT>(p, xv, pl, rseg_, gx_); if (PARK) { TZ_ROWS(k, r) gL[r] = gx_[k]; } }
    TZ_STAMP(PH_WARM_A);
    double viol = 0.0;
    TZ_ROWS(k, r) { const double hv = TZ_H(k, r); viol = fmax(viol, TZ_GX(k, r) - hv); sch = fmax(sch, fabs(hv)); }
    TZ_COLS(c, nz) scq = fmax(scq, fabs(qv[c]));
    tz_block_reduce3<RED_MAXU, RED_MAXU, RED_MAXU>(viol, scq, sch, red, rpar);      // also the scales of the stopping test
    TZ_STAMP(PH_WARM_B);
    // the stored start is ONE reference solve shared by all trajectories: a trajectory whose new rows it violates by more than
    // TZ_SEED_VIOL_MAX (equilibrated units) is far from the reference point and starts cold -- measured on jittered starts of the double
    // integrator (C oracle): below 0.1 the stored start needs 3-9 iterations, above 0.4 it needs 13-21, more than the 13 of a cold start
    if (seeded && viol > TZ_SEED_VIOL_MAX) { warm = false; TZ_ROWS(k, r) l_[k] = 1.0; }
    else {
    const double sig = fmin(fmax(pk.warm_floor, pk.warm_gain * viol), pk.warm_cap);
    const double sig2 = sig * sig;
    TZ_ROWS(k, r) {                  // slack >= sig, multiplier >= sig^2 / slack: onto the central path of mu = sig^2 where the pair was
      s_[k] = fmax(TZ_H(k, r) - TZ_GX(k, r), sig);           // below it; an inactive row keeps its multiplier ~ 0 instead of being
      l_[k] = fmax(l_[k], sig2 * tz_recip(s_[k]));           // lifted to sig (which alone set mu ~ sig * mean slack: 6 more iterations)
      vin[r] = l_[k];                                        // staged for the stopping test of the starting point (exact_rd at it == 0)
    }
    }
  }
  if (!warm) {
    was_shifted = 0;
    // ---- cold start: (P + G'G + reg) x = -q + G'h, then shift the slacks into the cone
    TZ_ROWS(k, r) vin[r] = 1.0;                          // w = 1 for the cold start point (the entries behind row mi stay zero for the whole launch)
    __syncthreads();
    if constexpr (TT) tz_gram_tt<TZ_TT_GU(MINW), TZ_TT_NST>(p, Hq, vin); else tz_gram(p, Hq, Pq, vin, kl);
    __syncthreads();
    TZ_ROWS(k, r) vin[r] = TZ_H(k, r);
    __syncthreads();
    if (!wave0) tz_ell_gemvT_part<TT>(p, vin, pl);
    __syncthreads();
    if (t < nzp) r1v[t] = (t < nz) ? tz_ell_colsum(pl, cseg) - qv[t] : 0.0;
    __syncthreads();
    if constexpr (TT) { okf = tz_cholesky_tt(p, Hq, dinv, dfac, flag); TZ_TT_AFTER_CHOL(p, Hq, dinv); TZ_TT_SOLVE(p, Hq, dinv, r1v, tmpz, xv); }
    else {
    if (p.chol1) { if (wave0) tz_cholesky_wave(p, Hq, dinv, flag); __syncthreads(); okf = (*flag == 0); }
    else okf = tz_cholesky(p, Hq, dinv, flag);
    if (p.chol1) tz_chol_solve_wave(p, Hq, dinv, r1v, xv); else tz_chol_solve(p, Hq, dinv, r1v, tmpz, xv);
    }
    tz_ell_gemv<MAXR, TT>(p, xv, pl, rseg_, gx_);
    if (PARK) { TZ_ROWS(k, r) gL[r] = gx_[k]; }
  }
  if (!warm) {
    double rmin = 1e300;
    scq = 0.0; sch = 0.0;
    TZ_ROWS(k, r) { const double hv = TZ_H(k, r); rmin = fmin(rmin, hv - TZ_GX(k, r)); sch = fmax(sch, fabs(hv)); }
    TZ_COLS(c, nz) scq = fmax(scq, fabs(qv[c]));
    tz_block_reduce3<RED_MIN, RED_MAX, RED_MAX>(rmin, scq, sch, red, rpar);
    const double shift = (rmin <= 1e-8) ? fmax(0.0, 1.0 - rmin) : 0.0;
    TZ_ROWS(k, r) s_[k] = TZ_H(k, r) - TZ_GX(k, r) + shift;
  }
  }
  // scales of the stopping test: parked in LDS (two spare slots of the reduction buffer) instead of four registers for the whole solve
  // (every thread stores the same two numbers -- the reductions above leave them in all threads -- so no barrier is needed for them:
  // a thread reads what its own wave wrote, and the first reader sits behind the barrier of the next reduction anyway)
  red[12] = 1.0 + scq; red[13] = 1.0 + sch;

  status = skip ? 3 : (okf ? 1 : 2);
  TZ_STAMP(PH_WARM);
  bool px_in_part = false;                  // `part` holds the partial sums of P x for the final x (left there by exact_rd)
  // diagonal-shift level of this solve (0: none), raised when a factorisation breaks down; the cold retry of a failed solve starts at
  // level 2 (1e-6): a breakdown the pivot test does not see (tiny positive pivots, a garbage step) is what made the first one fail
  // ... and, in the class with more than 64 variables, a warm-started step that follows a step with a breakdown takes that step's level (at most 2: 1e-6) BEFORE the
  // factorisation fails again, from the iteration on whose complementarity is within TZ_RLEV_GATE x of the one at which the previous step broke
  // down (earlier iterations stay unshifted: a shift from the first iteration on stalls the strictly convex two-input problem).  A
  // trajectory on a degenerate problem (an optimal face: the two-input 5-dim system) otherwise re-discovers the breakdown in every
  // step: three of its seven factorisations per step were such repeats.
  int rlev = retried ? 2 : 0;
  unsigned brk_now = 0;                      // high word of TZ_RLEV_GATE x mu at the first breakdown of this solve
  __builtin_amdgcn_s_setprio(0);
  for (it = 0; it < pk.max_iter && status == 1; ++it) {
    __builtin_amdgcn_s_setprio(TZ_PRIO_ELEM);
    TZ_FRESH_T();
    // the tolerances and step-rule constants are read from the kernel-argument segment where they are used (scalar loads) instead
    // of living in -- and being spilled from -- two dozen scalar registers for the whole kernel
    TzKargPtr kpi = (TzKargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kpi));
    const IpmParams& pi = *(const IpmParams*)kpi;
    TZ_STAMP(PH_STEP);
    double w_[MAXR], rp_[MAXR], ds_[MAXR], dl_[MAXR], g_[MAXR], is_[MAXR], il_[MAXR];      // scratch of this iteration only
#pragma unroll
    for (int k = 0; k < MAXR; ++k) { w_[k] = 0.0; rp_[k] = 0.0; ds_[k] = 0.0; dl_[k] = 0.0; g_[k] = 0.0; is_[k] = 1.0; il_[k] = 1.0; }
    // primal residual rp = G x + s - h and complementarity every iteration; the dual residual rd = P x + q + G'lambda is not
    // carried along at all: the right-hand sides below are written without it, and it is evaluated (exactly) only when rp and
    // mu already pass the test
    double nrp = 0, sl = 0, z0 = 0;
    TZ_ROWS(k, r) { rp_[k] = TZ_GX(k, r) + s_[k] - TZ_H(k, r); nrp = fmax(nrp, fabs(rp_[k])); sl += s_[k] * l_[k]; }
    TZ_STAMP(PH_T1);
    tz_block_reduce3<RED_MAXU, RED_SUM, RED_SUM, 2>(nrp, sl, z0, red, rpar);
    TZ_STAMP(PH_T2);
    const double mu = sl * pi.inv_mi;
    nrp *= tz_recip(red[13]);
    if (!(mu == mu) || !(nrp == nrp) || mu > 1e200) { status = 2; break; }
    TZ_STAMP(PH_T3);
    const bool fresh_warm = (it == 0) && warm;     // a warm start may BEGIN below mu_floor: its first Newton step is what removes the residuals
    if ((nrp <= pi.tol_res && mu <= pi.mu_tol) || (mu <= pi.mu_floor && !fresh_warm)) {
      const double e1 = exact_rd(fresh_warm);
      const double nrd = e1 * tz_recip(red[12]);
      TZ_STAMP(PH_GEMVT);
      if (!(nrd == nrd)) { status = 2; break; }
      if (nrd <= pi.tol_res && nrp <= pi.tol_res && mu <= pi.mu_tol) { status = 0; px_in_part = true; break; }
      if (mu <= pi.mu_floor && !fresh_warm) { status = (nrd <= pi.tol_loose && nrp <= pi.tol_loose) ? 0 : 2; px_in_part = true; break; }   // mu collapsed before the residuals: numerical
    }
    // Newton matrix.  is = 1/s, il = 1/lambda are the only two divisions per row and iteration.
    TZ_ROWS(k, r) { is_[k] = tz_recip(s_[k]); il_[k] = tz_recip(l_[k]); w_[k] = l_[k] * is_[k]; vin[r] = w_[k]; }
    __syncthreads();
    TZ_STAMP(PH_TOP);
    if constexpr (TT) {                      // (the problems with at most 64 variables have shown no breakdowns: the bookkeeping costs them 0.9 %)
      if (rlev == 0 && rlev_carry != 0 && warm && (unsigned)__builtin_amdgcn_readfirstlane((int)tz_hi(mu)) <= brk_hi)
        rlev = rlev_carry < TZ_RLEV_CARRY_MAX ? rlev_carry : TZ_RLEV_CARRY_MAX;
    }
    bool okc;
    bool have_y = false;                                // tmpz holds y = inv(L) r1 (forward substitution done while factoring)
    // A factorisation that breaks down (degenerate problems late in the solve: the weights of active and inactive rows are 1e18
    // apart and H loses definiteness in rounding) raises the diagonal-shift level -- 1 .. 4 = 1e-9, 1e-6, 1e-3, 1, kept for the
    // rest of the solve -- and the iteration is repeated from the same point (it counts as an iteration).  A shifted H only damps
    // the Newton step; the residuals are always exact.  `rlev` is uniform (scalar register); level 0, the normal case, costs one
    // scalar branch.
    __builtin_amdgcn_s_setprio(0);
    if constexpr (TT) tz_gram_tt<TZ_TT_GU(MINW), TZ_TT_NST>(p, Hq, vin, (PROF && t == 0) ? acc_ph : nullptr); else tz_gram(p, Hq, Pq, vin, kl, (PROF && t == 0) ? acc_ph : nullptr);
    __syncthreads();
    TZ_STAMP(PH_FORM);
    TZ_FRESH_T();
    if (rlev != 0) {
      const double rx = (rlev == 1) ? 1e-9 : (rlev == 2) ? 1e-6 : (rlev == 3) ? 1e-3 : 1.0;
      TZ_COLS(c, nz) {
        const int I = c >> 2, i = c & 3;
        Hq[TT ? (size_t)(((I * (I + 1)) >> 1) + I) * p.TS + 5 * i : (size_t)tz_hidx(c, c)] += rx;
      }
    }
    // ---- predictor (rc = s*lam):  H dx = -(P x + q) - G'(w rp).  The factorisation of H and the two products on the right
    // are independent: with chol1 wave 0 factors while waves 1-3 form the right-hand side.
    TZ_ROWS(k, r) vin[r] = w_[k] * rp_[k];
    if (t == 0) { flag[2] = 0; flag[3] = 0; }        // columns factored / waves done with the right-hand side
    __syncthreads();
    if (!TT && p.chol1) {
      if (wave0) {
        __builtin_amdgcn_s_setprio(TZ_PRIO);                 // the serial stretch of this workgroup: ahead of the co-resident waves
        tz_cholesky_wave(p, Hq, dinv, flag, flag + 2);
        __builtin_amdgcn_s_setprio(0);
      } else {
        tz_ell_gemvT_part<TT>(p, vin, pl);
        tz_gemvT_partial<NCG, 1, 3>(p.P, p.nP, nzp, xv, part2);
        // the three waves meet on a counter (wave 0 is busy factoring); wave 1 then assembles the right-hand side and runs the
        // forward substitution one tile column behind the factorisation
        tz_wave_sync();
        if ((tz_tid() & 63) == 0) __hip_atomic_fetch_add(flag + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (wave1) {
          bool ok1 = tz_spin_until(flag + 3, 3);
          const int c = tz_tid() & 63;
          double rv = 0.0;
          if (c < nzp) {
            const double pxq = (c < nz) ? tz_gemvT_get3(part2, nzp, c) + qv[c] : 0.0;
            rdv[c] = pxq;                                                 // P x + q, used again by the corrector
            rv = (c < nz) ? -pxq - tz_ell_colsum(pl, cseg) : 0.0;
          }
          ok1 = tz_fwd_trailing(p, Hq, dinv, rv, flag + 2, tmpz) && ok1;
          if (!ok1 && c == 0) flag[0] = 3;
        }
      }
      __syncthreads();
      okc = (*flag == 0);
      have_y = true;
      TZ_STAMP(PH_CHOL);
    } else {
      if (wave0) tz_gemvT_partial<NCG, 0, 1>(p.P, p.nP, nzp, xv, part2);
      else tz_ell_gemvT_part<TT>(p, vin, pl);
      __syncthreads();
      TZ_COLS(c, nzp) {
        const double pxq = (c < nz) ? part2[c] + qv[c] : 0.0;
        rdv[c] = pxq;
        r1v[c] = (c < nz) ? -pxq - tz_ell_colsum(pl, cseg) : 0.0;
      }
      __syncthreads();
      TZ_STAMP(PH_GEMVT);
      if constexpr (TT) { okc = tz_cholesky_tt(p, Hq, dinv, dfac, flag, (PROF && t == 0) ? acc_ph : nullptr); TZ_TT_AFTER_CHOL(p, Hq, dinv); }
      else okc = tz_cholesky(p, Hq, dinv, flag, (PROF && t == 0) ? acc_ph : nullptr);
      TZ_STAMP(PH_CHOL);
    }
    if (!okc) {
      if (rlev >= 4) { status = 2; break; }
      if constexpr (TT) { if (brk_now == 0) brk_now = (unsigned)__builtin_amdgcn_readfirstlane((int)tz_hi(mu * TZ_RLEV_GATE)); }
      rlev = __builtin_amdgcn_readfirstlane(rlev + 1);
      __syncthreads();                                 // every thread has read the failure flag
      if (t == 0) flag[0] = 0;
      __syncthreads();
      continue;
    }
    TZ_FRESH_T();
    if constexpr (TT) TZ_TT_SOLVE(p, Hq, dinv, r1v, tmpz, dxv);
    else if (p.chol1) tz_chol_solve_wave(p, Hq, dinv, have_y ? tmpz : r1v, dxv, have_y); else tz_chol_solve(p, Hq, dinv, r1v, tmpz, dxv);
    __syncthreads();
    TZ_STAMP(PH_SOLVE);
    tz_ell_gemv<MAXR, TT>(p, dxv, pl, rseg_, g_);
    TZ_STAMP(PH_GEMV);
    __builtin_amdgcn_s_setprio(TZ_PRIO_ELEM);
    // step to the boundary: alpha = 1 / max(1, max_i(-dv_i / v_i))
    TZ_FRESH_T();
    double mp = 0.0, md = 0.0, z4 = 0;
    TZ_ROWS(k, r) {
      const double ds = -rp_[k] - g_[k];
      const double dl = -l_[k] - w_[k] * ds;
      ds_[k] = ds; dl_[k] = dl;
      mp = fmax(mp, -ds * is_[k]);
      md = fmax(md, -dl * il_[k]);
      z4 += (s_[k] + ds) * (l_[k] + dl);
    }
    // z4: complementarity after the affine step, summed for the full step in the same reduction as the step lengths: when the full
    // step is feasible (the usual case from a warm start) that sum is the one wanted, otherwise it is formed again with ap, ad
    tz_block_reduce3<RED_MAXU, RED_MAXU, RED_SUM>(mp, md, z4, red, rpar);
    const double ap = tz_recip(fmax(1.0, mp)), ad = tz_recip(fmax(1.0, md));
    double muaff = z4;
    if (mp > 1.0 || md > 1.0) {
      double z1 = 0, z2 = 0;
      muaff = 0.0;
      TZ_ROWS(k, r) muaff += (s_[k] + ap * ds_[k]) * (l_[k] + ad * dl_[k]);
      tz_block_reduce3<RED_SUM, RED_SUM, RED_SUM, 1>(muaff, z1, z2, red, rpar);
    }
    muaff *= pi.inv_mi;
    const double sfr = (attempt == 0) ? pi.step_frac : pi.step_frac_retry;
    if (fmin(ap, ad) >= pi.aff_thr && muaff <= pi.aff_mu * mu) {
      // the Newton (predictor) step is already (almost) a full step and kills complementarity: take it, skip the corrector
      const double mmA = fmax(mp, md);
      const double alphaA = (mmA > sfr) ? sfr * tz_recip(mmA) : 1.0;
      TZ_COLS(c, nz) xv[c] += alphaA * dxv[c];
      TZ_ROWS(k, r) { s_[k] += alphaA * ds_[k]; l_[k] += alphaA * dl_[k]; TZ_ADD_GX(k, r, alphaA * g_[k]); }
      __syncthreads();
      continue;
    }
    double sigma = muaff * tz_recip(mu); sigma = sigma * sigma * sigma;
    TZ_FRESH_T();
    // ---- corrector (rc = s*lam + dsa*dla - sigma mu):  H dx = -(P x + q) - G'(lam + (lam rp - rc) / s) -------------------
    TZ_ROWS(k, r) {
      const double rc = s_[k] * l_[k] + ds_[k] * dl_[k] - sigma * mu;
      vin[r] = l_[k] + (l_[k] * rp_[k] - rc) * is_[k];
      ds_[k] = rc;                                  // keep rc for the dl formula
    }
    __syncthreads();
    TZ_STAMP(PH_ELEM);
    __builtin_amdgcn_s_setprio(0);
    if (!wave0) tz_ell_gemvT_part<TT>(p, vin, pl);
    __syncthreads();
    TZ_COLS(c, nzp) r1v[c] = (c < nz) ? -rdv[c] - tz_ell_colsum(pl, cseg) : 0.0;
    __syncthreads();
    TZ_STAMP(PH_GEMVT);
    if constexpr (TT) TZ_TT_SOLVE(p, Hq, dinv, r1v, tmpz, dxv);
    else if (p.chol1) tz_chol_solve_wave(p, Hq, dinv, r1v, dxv); else tz_chol_solve(p, Hq, dinv, r1v, tmpz, dxv);
    __syncthreads();
    TZ_STAMP(PH_SOLVE);
    tz_ell_gemv<MAXR, TT>(p, dxv, pl, rseg_, g_);
    TZ_STAMP(PH_GEMV);
    __builtin_amdgcn_s_setprio(TZ_PRIO_ELEM);
    double ms = 0.0, ml = 0.0, z3 = 0;
    TZ_ROWS(k, r) {
      const double rc = ds_[k];
      const double ds = -rp_[k] - g_[k];
      const double dl = (-rc - l_[k] * ds) * is_[k];
      ds_[k] = ds; dl_[k] = dl;
      ms = fmax(ms, -ds * is_[k]);
      ml = fmax(ml, -dl * il_[k]);
    }
    tz_block_reduce3<RED_MAXU, RED_MAXU, RED_SUM, 2>(ms, ml, z3, red, rpar);
    const double mm = fmax(ms, ml);
    const double alpha = (mm * 1.0 > sfr) ? sfr * tz_recip(mm) : 1.0;      // min(1, sfr * min_i(-v_i/dv_i))
    TZ_COLS(c, nz) xv[c] += alpha * dxv[c];
    TZ_ROWS(k, r) { s_[k] += alpha * ds_[k]; l_[k] += alpha * dl_[k]; TZ_ADD_GX(k, r, alpha * g_[k]); }
    __syncthreads();
  }
  __builtin_amdgcn_s_setprio(TZ_PRIO_GLUE);       // stopping test done: recovery, plant update, tube, maps and warm start of the next step are short
  TZ_FRESH_T();
  work_f = __builtin_amdgcn_readfirstlane(work_f + it + ((warm || skip) ? 0 : 1));
  if (TT && status == 0 && !skip && attempt == 0) {
    if (brk_now != 0) { rlev_carry = __builtin_amdgcn_readfirstlane(rlev); brk_hi = brk_now; }
    else if (rlev == 0) { rlev_carry = 0; brk_hi = 0; }               // a step that needed no shift at all: forget
  }
  if (status != 0 && !skip && attempt == 0) {
    attempt = 1; src = 0;
    __syncthreads();
    if (t == 0) flag[0] = 0;
    TZ_COLS(c, nzp) xv[c] = 0.0;
    TZ_ROWS(k, r) { s_[k] = 1.0; l_[k] = 1.0; }
    __syncthreads();
    goto retry_solve;
  }
  if (status != 0 && !skip) {
    // Both attempts failed.  Only a certificate of primal infeasibility carried by the multipliers -- y = lambda / max(lambda) >= 0
    // with G'y ~ 0 and h'y < 0 (Farkas) -- makes it TZ_INFEASIBLE (the reference raises 'Problem is unbounded' for an infeasible
    // problem, tzddpc/tzddpc.py:374-375); every other failure stays max-iter / numerical.  The failed iterate is not used:
    // x = 0, i.e. u = K e and the nominal state follows Phi, exactly like a step whose parameter rows are violated.
    double lm = 0.0, z1 = 0.0, z2 = 0.0;
    TZ_ROWS(k, r) lm = fmax(lm, l_[k]);
    tz_block_reduce3<RED_MAX, RED_MAX, RED_MAX, 1>(lm, z1, z2, red, rpar);
    const bool lm_ok = lm > 0.0 && lm < 1e300;
    const double il = lm_ok ? 1.0 / lm : 0.0;
    double hy = 0.0;
    TZ_ROWS(k, r) { const double y = lm_ok ? l_[k] * il : 0.0; vin[r] = y; hy += TZ_H(k, r) * y; }
    __syncthreads();
    if (!wave0) tz_ell_gemvT_part<TT>(p, vin, pl);
    __syncthreads();
    double gmax = fabs(tz_ell_colsum(pl, cseg));
    tz_block_reduce3<RED_MAX, RED_SUM, RED_SUM, 2>(gmax, hy, z2, red, rpar);
    if (lm_ok && hy < -1e-6 && gmax <= 1e-6 * fmax(1.0, -hy)) status = 3;
    TZ_COLS(c, nzp) xv[c] = 0.0;
    px_in_part = false;
    __syncthreads();
  }
  work_s += 1;
  TzKargPtr kpe = (TzKargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kpe));
  const IpmParams& pe = *(const IpmParams*)kpe;
  if (step == nsteps - 1) {                 // what a later launch (or the host) reads: solution, multipliers, status
    TZ_COLS(c, nz) pe.x[(size_t)b * nz + c] = xv[c];
    TZ_ROWS(k, r) { pe.s[(size_t)b * mi + r] = s_[k]; pe.lam[(size_t)b * mi + r] = l_[k]; }
    if (t == 0) {
      pe.status[b] = status; pe.iters[b] = it;
      pe.shift_state[b] = was_shifted;             // (also without a shift policy: a stored start's mark -2 must not survive the launch)
      if (pe.status_copy) pe.status_copy[b] = status;
      if (pe.work) { atomicAdd(pe.work, (unsigned long long)work_f); atomicAdd(pe.work + 1, (unsigned long long)work_s); atomicMax(pe.work + 2, (unsigned long long)work_f); }
    }
  }
  TZ_STAMP(PH_ELEM);
  if (fused) {
    TzKargPtr kp1 = (TzKargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp1));
    const FuseParams& F = ((const IpmParams*)kp1)->F;
    // ---- recovery (tz_finish_kernel) and plant / error update (tz_plant_kernel) of this trajectory ----------------
    const int n = F.fin.n, m = F.fin.m, N = F.fin.N, nv = N * m;
    const double* x0 = stl + n;                         // nominal state this step started from
    const double* ec = tbl + (F.tube.pmax + 1) * n * n + (F.tube.pmax > 0 ? F.tube.pmax : 1) * (n + m) * n;   // LDS constants, see above
    const double *cA = ec, *cB = ec + n * n, *cK = cB + n * m, *cr1 = cK + n * m, *cR2 = cr1 + n, *cPhi = cR2 + n * n, *cGam = cPhi + n * n, *cDz = cGam + n * nv;
    const bool want_cost = F.cost_step != 0 || step == nsteps - 1;      // a cost that the next step overwrites is not formed
    if (F.lean_epilogue != 0 && !want_cost && !F.fin.v && !F.fin.xbar) {
      // a step inside a multi-step launch that reports neither cost nor v / xbar: everything the next step needs -- u = K e + v[0],
      // x+ = A x + B u + w, xbar+ = Phi_1 xbar + Gam_1 v[0], e+ -- is n (n + m) products: one wave, no workgroup barrier inside
      if (skip) __syncthreads();                         // x was zeroed after the last barrier
      if (t == 0 && status != 0 && F.plant.sticky && F.plant.sticky[b] == 0) F.plant.sticky[b] = status;
      if (t < 64) {
        const PlantParams& Q = F.plant;
        double xn = 0.0, xb = 0.0;
        if (t < n) {
          xn = stl[3 * TZ_NMAX + t];                          // w of this step (fetched in the prologue)
          for (int j = 0; j < n; ++j) { xn += cA[t * n + j] * stl[j]; xb += cPhi[t * n + j] * x0[j]; }
          for (int j = 0; j < m; ++j) {
            const double v0 = cDz[j] * xv[F.fin.vpos ? F.fin.vpos[j] : j];
            double u = v0;
            for (int i = 0; i < n; ++i) u += cK[j * n + i] * stl[2 * n + i];
            xn += cB[t * m + j] * u; xb += cGam[t * nv + j] * v0;
            if (t == 0 && Q.u_out) Q.u_out[(size_t)b * Q.u_stride + (size_t)step * F.u_step + j] = u;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // every lane has read the old state before any lane overwrites it
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t < n) {
          stl[t] = xn; stl[n + t] = xb; stl[2 * n + t] = xn - xb;
          if (Q.x_out) Q.x_out[(size_t)b * Q.x_stride + (size_t)step * F.x_step + t] = xn;
        }
      }
      __syncthreads();                          // the next step's tube pass reads the state
      TZ_STAMP(PH_EPILOGUE);
      continue;
    }
    __syncthreads();
    if (want_cost && !px_in_part) tz_gemvT_partial<NCG>(p.P, p.nP, nzp, xv, part);
    if (F.fin.recy) {                                    // equality rows eliminated by the host: v is an affine map of (xbar0, x)
      for (int c = t; c < nv; c += TZ_THREADS) {
        double a = F.fin.rec0[c];
        for (int j = 0; j < n; ++j) a += F.fin.recx[(size_t)c * n + j] * x0[j];
        for (int k = 0; k < nz; ++k) a += F.fin.recy[(size_t)c * nz + k] * xv[k];
        dxv[c] = a;
      }
    } else
    for (int c = t; c < nv; c += TZ_THREADS) dxv[c] = cDz[c] * xv[F.fin.vpos ? F.fin.vpos[c] : c];     // v in the caller's order
    __syncthreads();
    TZ_STAMP(PH_EPI_A);
    double acc = 0.0, z1 = 0.0, z2 = 0.0;
    if (want_cost) {
      TZ_COLS(c, nz) acc += xv[c] * (0.5 * (px_in_part ? part[c] : tz_gemvT_get(part, nzp, c)) + qv[c]);
      tz_block_reduce3<RED_SUM, RED_SUM, RED_SUM, 1>(acc, z1, z2, red, rpar);
    }
    if (t == 0 && !want_cost) { if (status != 0 && F.plant.sticky && F.plant.sticky[b] == 0) F.plant.sticky[b] = status; }
    if (t == 0 && want_cost) {
      double r = F.fin.r0;
      for (int i = 0; i < n; ++i) {
        r += cr1[i] * x0[i];
        for (int j = 0; j < n; ++j) r += x0[i] * cR2[i * n + j] * x0[j];
      }
      F.fin.cost[(size_t)b * F.fin.cost_stride + (size_t)step * F.cost_step] = (status == 0) ? acc / F.fin.cost_scale + r : INFINITY;
      if (status != 0 && F.plant.sticky && F.plant.sticky[b] == 0) F.plant.sticky[b] = status;
    }
    if (F.fin.v) for (int c = t; c < nv; c += TZ_THREADS) F.fin.v[(size_t)b * nv + c] = dxv[c];
    if (F.fin.xbar) {                                    // whole predicted nominal trajectory (single-step launches)
      for (int r = t; r < (N + 1) * n; r += TZ_THREADS) {
        double a = 0.0;
        for (int j = 0; j < n; ++j) a += F.fin.Phi[(size_t)r * n + j] * x0[j];
        const double* g = F.fin.Gam + (size_t)r * nv;
        for (int c = 0; c < nv; ++c) a += g[c] * dxv[c];
        F.fin.xbar[(size_t)b * (N + 1) * n + r] = a;
      }
    }
    {                                                    // xbar[1], the next nominal state: component i by wave i mod 4, the sums folded across its lanes
      const int ln = t & 63;
      for (int i = __builtin_amdgcn_readfirstlane(t >> 6); i < n; i += TZ_NWAVES) {
        double a = (ln < n) ? cPhi[i * n + ln] * x0[ln] : 0.0;
        for (int c = ln; c < nv; c += 64) a += cGam[i * nv + c] * dxv[c];
        a = tz_wave_reduce<RED_SUM>(a);
        if (ln == 0) tmpz[i] = a;
      }
    }
    __syncthreads();
    TZ_STAMP(PH_EPI_B);
    if (t < 64) {
      const PlantParams& Q = F.plant;
      double xn = 0.0, xb = 0.0;
      if (t < n) {
        xn = stl[3 * TZ_NMAX + t];                          // w of this step (fetched in the prologue)
        for (int j = 0; j < n; ++j) xn += cA[t * n + j] * stl[j];
        for (int j = 0; j < m; ++j) {
          double u = dxv[j];
          for (int i = 0; i < n; ++i) u += cK[j * n + i] * stl[2 * n + i];
          xn += cB[t * m + j] * u;
          if (t == 0 && Q.u_out) Q.u_out[(size_t)b * Q.u_stride + (size_t)step * F.u_step + j] = u;
        }
        xb = tmpz[t];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                       // every lane has read the old state before any lane overwrites it
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (t < n) {
        stl[t] = xn; stl[n + t] = xb; stl[2 * n + t] = xn - xb;
        if (step == nsteps - 1) { Q.x[(size_t)b * n + t] = xn; Q.xbar[(size_t)b * n + t] = xb; Q.e[(size_t)b * n + t] = xn - xb; }
        if (Q.x_out) Q.x_out[(size_t)b * Q.x_stride + (size_t)step * F.x_step + t] = xn;
      }
    }
    __syncthreads();                          // the next step's tube pass reads the state
    TZ_STAMP(PH_EPILOGUE);
  }
  }   // steps
  if (PROF && t == 0) {
    TZ_STAMP(PH_EPILOGUE);
    acc_ph[PH_TOTAL] = tprev - tstart; acc_ph[7] = (unsigned long long)it;
    for (int i = 0; i < PH_COUNT; ++i) p.prof[i] = acc_ph[i];
  }
#undef TZ_STAMP
#undef TZ_ROWS
#undef TZ_COLS
#undef TZ_FRESH_T
}
