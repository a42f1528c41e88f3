// C-ABI of the MI355X TZDDPC hot path (see include/tzddpc.h).  Host side: uploads the problem,
// packs the constraint matrix into 4x4 MFMA patches, builds the static Gram work plan, launches the
// kernels of tz_kernels.hip.h on one HIP stream.  No torch types, no CPU compute fallback: every
// numeric result comes out of a kernel.
#include "tz_kernels.hip.h"

// Experiment switches (tools/*.sh, tools/*.py) are read from the environment by the DIAGNOSTIC build only (libtzddpc_hip_prof.so,
// -DTZ_PROFILE=1); the release library reads no environment variable: what a caller may choose goes through tz_problem_desc
// (plan_flags) and the tz_problem_set_* entry points.
static inline const char* tz_dev_getenv(const char* name) {
#if TZ_PROFILE
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
#include "../../include/tzddpc.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

static thread_local std::string g_err;

#define TZ_FAIL(code, ...) do { char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__); g_err = _b; return (code); } while (0)
#define TZ_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { \
    char _b[512]; snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    g_err = _b; return TZ_ERR_HIP; } } while (0)

namespace {

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t count) {
    if (p) { (void)hipFree(p); p = nullptr; }
    n = count;
    if (count == 0) return hipSuccess;
    return hipMalloc((void**)&p, count * sizeof(T));
  }
  hipError_t upload(const T* src, size_t count) {
    hipError_t e = alloc(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
  }
  hipError_t upload(const std::vector<T>& v) { return upload(v.data(), v.size()); }
  void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

// Balanced lane-ELL of a sparse matrix for the matrix-vector products of tz_ipm_kernel (TzEll in tz_ipm.hip.h).
// outs[o] = (index, value) pairs of output o; NL physical lanes per pass; the virtual lane count VL is a multiple of NL.
// compact: 8-byte values and 16-bit indices in two arrays (the tile-triangle class, see TzEll) instead of 16-byte records.
struct DevEll {
  DevBuf<TzEllEnt> ent; DevBuf<int> seg; DevBuf<double> val; DevBuf<unsigned short> idx;
  int L = 1, VL = 0;
  hipError_t build(const std::vector<std::vector<std::pair<int, double>>>& outs, int NL, int VLwant, bool compact) {
    VL = ((std::max(VLwant, 1) + NL - 1) / NL) * NL;
    size_t longest = 1;
    for (auto& o : outs) longest = std::max(longest, o.size());
    for (L = 1; L <= (int)longest; ++L) {
      size_t lanes = 0;
      for (auto& o : outs) lanes += (o.size() + L - 1) / L;
      if (lanes <= (size_t)VL) break;
    }
    std::vector<TzEllEnt> v(compact ? 0 : (size_t)VL * L, TzEllEnt{0.0, 0u, 0u});
    std::vector<double> cv(compact ? (size_t)VL * L : 0, 0.0);
    std::vector<unsigned short> ci(compact ? (size_t)VL * L : 0, (unsigned short)0);
    std::vector<int> sg(std::max<size_t>(outs.size(), 1), 0);
    int lane = 0;
    for (size_t o = 0; o < outs.size(); ++o) {
      const int cnt = (int)((outs[o].size() + L - 1) / L);
      sg[o] = lane | (cnt << 16);
      for (size_t e = 0; e < outs[o].size(); ++e) {
        const int vl = lane + (int)(e / L), slot = (int)(e % L);
        const size_t pos = ((size_t)(vl / NL) * L + slot) * NL + (vl % NL);
        if (compact) { cv[pos] = outs[o][e].second; ci[pos] = (unsigned short)outs[o][e].first; }
        else v[pos] = TzEllEnt{outs[o][e].second, (unsigned)outs[o][e].first * 8u, 0u};
      }
      lane += cnt;
    }
    hipError_t e;
    if (compact) {
      if ((e = val.upload(cv)) != hipSuccess) return e;
      if ((e = idx.upload(ci)) != hipSuccess) return e;
    } else if ((e = ent.upload(v)) != hipSuccess) return e;
    return seg.upload(sg);
  }
  TzEll view() const { return TzEll{L, VL, ent.p, seg.p, val.p, idx.p}; }
  void swap(DevEll& o) { ent.swap(o.ent); seg.swap(o.seg); val.swap(o.val); idx.swap(o.idx); std::swap(L, o.L); std::swap(VL, o.VL); }
};

struct DevCsr {          // device copy of a tz_affmap (CSR in the ABI) re-laid out as ELL, see TzCsr
  DevBuf<TzEllEnt> ent;
  DevBuf<double> c0;
  int rows = 0, W = 1;
  // perm (may be null): device row i is row perm[i] of the map
  hipError_t upload(const tz_affmap& m, const int* perm = nullptr) {
    rows = m.rows; W = 1;
    for (int r = 0; r < m.rows; ++r) W = std::max(W, m.ptr[r + 1] - m.ptr[r]);
    std::vector<TzEllEnt> ve((size_t)W * std::max(rows, 1), TzEllEnt{0.0, 0u, 0u});
    std::vector<double> cc((size_t)std::max(rows, 1), 0.0);
    for (int i = 0; i < m.rows; ++i) {
      const int r = perm ? perm[i] : i;
      cc[i] = m.c0[r];
      for (int e = m.ptr[r]; e < m.ptr[r + 1]; ++e) ve[(size_t)(e - m.ptr[r]) * rows + i] = TzEllEnt{m.val[e], (unsigned)m.col[e] * 8u, 0u};
    }
    hipError_t e;
    if ((e = ent.upload(ve)) != hipSuccess) return e;
    return c0.upload(cc.data(), (size_t)m.rows);
  }
  TzCsr view() const { return TzCsr{rows, W, ent.p, c0.p}; }
};

enum { K_TUBE = 0, K_IPM = 1, K_FINISH = 2, K_PLANT = 3, K_COUNT = 4 };

typedef void (*ipm_fn_t)(IpmParams);
// Kernel variants: <rows per thread, 64-column groups, workgroups per CU the register budget is compiled for>.
//   nz <= 64   (one column group): quad layout, single-wave Cholesky; always fits four workgroups per CU (128 registers).
//   nz  > 64   tile-triangle layout, blocked Gram, two-phase Cholesky (tz_tt.hip.h): 256 registers (two workgroups per CU) or
//              512 (one) -- the blocked Gram keeps an 8 x 8 block of tiles in accumulators.
//   more than 1024 rows (5 or 6 per thread): tile-triangle variants only.
template <int R> ipm_fn_t ipm_pick_tt(int ncg, int w) {
  if (w >= 2) { switch (ncg) { case 1: case 2: return tz_ipm_kernel<R, 2, 2>; case 3: return tz_ipm_kernel<R, 3, 2>; default: return tz_ipm_kernel<R, 4, 2>; } }
  switch (ncg) { case 1: case 2: return tz_ipm_kernel<R, 2, 1>; case 3: return tz_ipm_kernel<R, 3, 1>; default: return tz_ipm_kernel<R, 4, 1>; }
}
// tt (out): the variant uses the tile-triangle layout
ipm_fn_t ipm_kernel_for(int maxr, int ncg, int wgs_per_cu, bool& tt) {
#ifdef TZ_ONLY_SMALL      // development builds (assembly study, quick A/B of the bench problem): only the variant for mi <= 256, nz <= 64
  tt = false;
  return (maxr == 1 && ncg == 1) ? tz_ipm_kernel<1, 1, TZ_MINWAVES> : nullptr;
#elif defined(TZ_ONLY_TT)  // development builds of the tile-triangle class: -DTZ_ONLY_TT=R,NCG,W  (one variant, seconds to build)
  tt = true; (void)wgs_per_cu;
  return tz_ipm_kernel<TZ_ONLY_TT>;
#else
  tt = (ncg >= 2 || maxr > 4);
  if (!tt) {
    switch (maxr) { case 1: return tz_ipm_kernel<1, 1, TZ_MINWAVES>; case 2: return tz_ipm_kernel<2, 1, TZ_MINWAVES>; case 3: return tz_ipm_kernel<3, 1, TZ_MINWAVES>; default: return tz_ipm_kernel<4, 1, TZ_MINWAVES>; }
  }
  switch (maxr) {
    case 1: return ipm_pick_tt<1>(ncg, wgs_per_cu); case 2: return ipm_pick_tt<2>(ncg, wgs_per_cu); case 3: return ipm_pick_tt<3>(ncg, wgs_per_cu);
    case 4: return ipm_pick_tt<4>(ncg, wgs_per_cu);
    case 5: return ncg <= 3 ? tz_ipm_kernel<5, 3, 1> : tz_ipm_kernel<5, 4, 1>;
    default: return ncg <= 3 ? tz_ipm_kernel<6, 3, 1> : tz_ipm_kernel<6, 4, 1>;
  }
#endif
}

}  // namespace

struct tz_problem {
  int device = 0;
  int n = 0, m = 0, N = 0, nz = 0, mi = 0, ntheta = 0, npar = 0, nc_rows = 0, pmax = 0;
  int nzp = 0, mip = 0, Tz = 0, Kc = 0, nquads = 0, nklist = 0, nP = 0;
  int max_iter = 40;
  double tol = 1e-10, reg = 1e-12, step_frac = 0.99, cost_scale = 1.0, r0 = 0.0;
  // constants
  DevBuf<double> P, G, Gt, Gp, act_scale, Dz, Phi, Gam, r1, R2, CK, DK, K, CKpow, Ttube, par_lo, par_hi, rec0, recx, recy;
  DevBuf<int> power, row_of, klist, item_ptr, smask, shift_var, shift_row;
  DevBuf<double> shift_xs, shift_ls;
  int shift_policy = 0;        // 0 never, 1 always, k >= 2: after a step of >= k iterations (tz_problem_set_warm_shift)
  int shift_quiet = 16;        // leave the shifted regime after this many one-iteration shifted steps (0: never)
  bool have_shift = false;
  DevBuf<IpmItem> items;
  DevCsr q, h, par;
  DevEll eg, et;
  int nell = 0;
  size_t lds_bytes = 0;
  int64_t mfma_gram = 0, mfma_chol = 0, mfma_issued = 0;
  // workspace (capacity Bcap)
  int Bcap = 0;
  DevBuf<double> theta, qv, hv, x, s, lam, v, xbar, cost, in_x0, in_e0;
  DevBuf<int> prestatus, status, iters, sticky, prev_status, shift_state;
  DevBuf<uint8_t> active;
  // closed-loop state / plant (simulate)
  DevBuf<double> st_x, st_xbar, st_e, plantA, plantB, noise, xtraj, utraj, costtraj;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool[K_COUNT];
  size_t ev_used[K_COUNT] = {0, 0, 0, 0};
  double t_ms[K_COUNT] = {0, 0, 0, 0};
  int64_t t_count[K_COUNT] = {0, 0, 0, 0};
  int lastB = 0;
  bool prof = false;
  bool have_prev = false; int prevB = 0;   // x / s / lambda of the previous closed-loop step are valid for prevB trajectories
  DevBuf<double> ref_x, ref_lam;          // stored start (tz_problem_store_start): solution and multipliers of one reference solve
  bool have_ref = false;
  double warm_floor = 1e-8, warm_gain = 1.0, warm_cap = 1e300, mu_factor = 1e-3, res_factor = 100.0, aff_thr = 0.99, aff_mu = 1e-3;
  bool warm_enabled = true;
  int ntube = 0;
  bool tt = false;             // tile-triangle layout / blocked Gram / two-phase Cholesky (nz > 64 or more than 1024 rows)
  int TS = 16, ntile = 0, gu = 0;
  size_t hsize = 0;            // doubles of factor storage in LDS
  DevBuf<TzGUnit> gunits; DevBuf<int> gunit_ptr;
  DevBuf<int> vpos;            // staircase ordering (tile-triangle class): device position of v[k, j]; null = identity
  std::vector<int> permc, permr;   // device variable / row i is the caller's permc[i] / permr[i] (empty = identity)
  bool chol1 = false;          // single-wave Cholesky overlapped with the predictor's G' product (Tz <= 16; TZ_CHOL1=0 disables)
  bool ksplit = false;         // Gram by k-split (Tz <= TZ_KS_TZ; TZ_KSPLIT=0 keeps the item plan)
  bool fuse_enabled = true;    // closed-loop steps in one launch (TZ_PLAN_UNFUSED: four kernels per step, same arithmetic)
  bool staircase = false;      // tile-triangle class: variables in time order, rows by last non-zero column (library-internal)
  bool toeplitz = false;       // G x / G'v as block-Toeplitz convolutions over the horizon (not built: see DESIGN.md)
  bool lean_epilogue = false;  // FuseParams::lean_epilogue
  struct tz_genstack* tube_stack = nullptr;   // literal problems: decision-independent generators, evaluated per solve (not owned)
  DevBuf<double> ts_zeta, ts_c, ts_rx, ts_ru;  int ts_cap = 0;
  int maxr = 1, ncg = 1;
  void (*ipm_fn)(IpmParams) = nullptr;
  DevBuf<unsigned long long> prof_buf, work_buf;
};

extern "C" int tube_stack_theta(tz_problem* p, int B, const double* d_e0, hipStream_t st);   // literal problems: theta from the attached stack (defined with the K1g entry points)

namespace {

int ensure_workspace(tz_problem* p, int B) {
  if (B <= p->Bcap) return TZ_OK;
  TZ_HIP(hipSetDevice(p->device));
  size_t b = (size_t)B;
  TZ_HIP(p->theta.alloc(b * p->ntheta));
  TZ_HIP(p->qv.alloc(b * p->nz));
  TZ_HIP(p->hv.alloc(b * p->mi));
  TZ_HIP(p->x.alloc(b * p->nz));
  TZ_HIP(p->s.alloc(b * p->mi));
  TZ_HIP(p->lam.alloc(b * p->mi));
  TZ_HIP(p->v.alloc(b * p->N * p->m));
  TZ_HIP(p->xbar.alloc(b * (p->N + 1) * p->n));
  TZ_HIP(p->cost.alloc(b));
  TZ_HIP(p->in_x0.alloc(b * p->n));
  TZ_HIP(p->in_e0.alloc(b * p->n));
  TZ_HIP(p->prestatus.alloc(b));
  TZ_HIP(p->status.alloc(b));
  TZ_HIP(p->iters.alloc(b));
  TZ_HIP(p->sticky.alloc(b));
  TZ_HIP(p->prev_status.alloc(b));
  TZ_HIP(p->shift_state.alloc(b));
  TZ_HIP(hipMemset(p->shift_state.p, 0, b * sizeof(int)));
  TZ_HIP(p->active.alloc(b * std::max(p->nc_rows, 1)));
  TZ_HIP(p->st_x.alloc(b * p->n));
  TZ_HIP(p->st_xbar.alloc(b * p->n));
  TZ_HIP(p->st_e.alloc(b * p->n));
  p->Bcap = B;
  p->have_prev = false;
  return TZ_OK;
}

void drain_timing_one(tz_problem* p, int k) {
  for (size_t i = 0; i < p->ev_used[k]; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(p->ev_pool[k][i].second) == hipSuccess &&
        hipEventElapsedTime(&ms, p->ev_pool[k][i].first, p->ev_pool[k][i].second) == hipSuccess) {
      p->t_ms[k] += ms; p->t_count[k]++;
    }
  }
  p->ev_used[k] = 0;
}

struct Timer {
  tz_problem* p; int k; hipEvent_t e0 = nullptr, e1 = nullptr;
  Timer(tz_problem* p_, int k_) : p(p_), k(k_) {
    if (!p->timing) return;
    if (p->ev_used[k] == p->ev_pool[k].size()) {
      if (p->ev_pool[k].size() >= 4096) drain_timing_one(p, k);       // pool exhausted: fold what is recorded (one stream sync)
      else {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        p->ev_pool[k].push_back({a, b});
      }
    }
    e0 = p->ev_pool[k][p->ev_used[k]].first; e1 = p->ev_pool[k][p->ev_used[k]].second;
    p->ev_used[k]++;
    (void)hipEventRecord(e0, p->stream);
  }
  ~Timer() { if (e1) (void)hipEventRecord(e1, p->stream); }
};

void drain_timing(tz_problem* p) {
  for (int k = 0; k < K_COUNT; ++k) drain_timing_one(p, k);
}

IpmParams ipm_params(tz_problem* p, int B, int* d_status, int* d_iters, bool warm, bool track_prev) {
  IpmParams ip{};
  ip.B = B; ip.nz = p->nz; ip.mi = p->mi; ip.nzp = p->nzp; ip.mip = p->mip; ip.Tz = p->Tz; ip.Kc = p->Kc; ip.nquads = p->nquads;
  ip.P = p->P.p; ip.Gp = p->Gp.p; ip.items = p->items.p; ip.item_ptr = p->item_ptr.p; ip.klist = p->klist.p; ip.smask = p->smask.p; ip.eg = p->eg.view(); ip.et = p->et.view(); ip.nell = p->nell;
  ip.q = p->qv.p; ip.h = p->hv.p; ip.prestatus = p->prestatus.p; ip.x = p->x.p; ip.s = p->s.p; ip.lam = p->lam.p;
  ip.status = d_status; ip.iters = d_iters ? d_iters : p->iters.p;
  ip.max_iter = p->max_iter; ip.tol = p->tol; ip.reg = p->reg; ip.step_frac = p->step_frac; ip.mu_tol = p->tol * p->mu_factor; ip.tol_res = p->tol * p->res_factor;
  ip.inv_mi = 1.0 / p->mi; ip.mu_floor = 1e-3 * ip.mu_tol; ip.tol_loose = 1e3 * p->tol; ip.step_frac_retry = std::min(p->step_frac, 0.99);
  ip.prof = p->prof ? p->prof_buf.p : nullptr;
  ip.work = p->timing ? p->work_buf.p : nullptr;
  ip.TS = p->TS; ip.ntile = p->ntile; ip.gu = p->gu; ip.gunits = p->gunits.p; ip.gunit_ptr = p->gunit_ptr.p;
  ip.nklist = p->nklist; ip.nP = p->nP; ip.ksplit = p->ksplit ? 1 : 0; ip.chol1 = p->chol1 ? 1 : 0; ip.ntube = p->ntube;
  ip.shift_policy = p->have_shift ? p->shift_policy : 0; ip.shift_quiet = p->shift_quiet;
  ip.shift_state = p->shift_state.p;
  ip.sx = p->shift_var.p; ip.sr = p->shift_row.p; ip.sxs = p->shift_xs.p; ip.sls = p->shift_ls.p;
  ip.warm = warm ? 1 : 0; ip.warm_floor = p->warm_floor;
  ip.warm_gain = p->warm_gain; ip.warm_cap = p->warm_cap; ip.aff_thr = p->aff_thr; ip.aff_mu = p->aff_mu;
  ip.prev_status = warm ? p->prev_status.p : nullptr;
  ip.status_copy = track_prev ? p->prev_status.p : nullptr;
  ip.F.on = 0;
  return ip;
}

// One closed-loop step of B trajectories in ONE launch (tz_ipm_kernel with F.on): tube, parameter maps, interior point,
// recovery / objective and plant update; theta, q and h never reach HBM.  Same arithmetic as launch_solve + launch_plant.
struct StepStrides { size_t w, u, x, cost; };     // element offsets per closed-loop step inside one launch
int launch_step_fused(tz_problem* p, int B, double* d_x, double* d_xbar, double* d_e, const double* d_w, size_t w_stride,
                      const double* d_A, const double* d_Bm, double* d_u, size_t u_stride, double* d_xout, size_t x_stride,
                      double* d_cost, size_t cost_stride, int* d_status, int* d_sticky, bool warm,
                      int nsteps = 1, StepStrides ss = StepStrides{0, 0, 0, 0}, bool sticky_fresh = false) {
  p->lastB = B;
  Timer tm(p, K_IPM);
  IpmParams ip = ipm_params(p, B, d_status, p->iters.p, warm, true);
  FuseParams& F = ip.F;
  F.on = 1; F.npar = p->npar; F.ntheta = p->ntheta; F.lean_epilogue = p->lean_epilogue ? 1 : 0; F.sticky_fresh = sticky_fresh ? 1 : 0;
  F.nsteps = nsteps; F.warm_steps = p->warm_enabled ? 1 : 0; ip.warm_steps = F.warm_steps;
  F.w_step = ss.w; F.u_step = ss.u; F.x_step = ss.x; F.cost_step = ss.cost;
  F.tube = TubeParams{B, p->n, p->m, p->N, p->pmax, p->ntheta, p->CKpow.p, p->Ttube.p, p->power.p, d_xbar, d_e, nullptr, nullptr};
  F.qmap = p->q.view(); F.hmap = p->h.view(); F.parmap = p->par.view(); F.par_lo = p->par_lo.p; F.par_hi = p->par_hi.p;
  F.fin = FinishParams{B, p->n, p->m, p->N, p->nz, p->mi, p->nzp, p->nc_rows, p->P.p, p->Dz.p, p->Phi.p, p->Gam.p,
                       p->r1.p, p->R2.p, p->r0, p->cost_scale, p->row_of.p, p->act_scale.p, d_xbar, nullptr, nullptr, nullptr, nullptr,
                       d_status, nsteps > 1 ? nullptr : p->v.p, nsteps > 1 ? nullptr : p->xbar.p, d_cost, nullptr, cost_stride, p->vpos.p,
                       p->rec0.p, p->recx.p, p->recy.p};
  F.plant = PlantParams{B, p->n, p->m, p->N, p->K.p, d_A, d_Bm, nullptr, nullptr, d_w, w_stride, d_status, d_x, d_xbar, d_e,
                        d_u, u_stride, d_xout, x_stride, d_sticky};
  hipLaunchKernelGGL(p->ipm_fn, dim3(B), dim3(TZ_THREADS), p->lds_bytes, p->stream, ip);
  TZ_HIP(hipGetLastError());
  return TZ_OK;
}

// Core launch sequence on device-resident inputs: tube -> affine -> ipm -> finish.
int launch_solve(tz_problem* p, int B, const double* d_xbar0, const double* d_e0,
                 double* d_v, double* d_xbar, double* d_cost, int* d_status, int* d_iters, uint8_t* d_active, size_t cost_stride = 1, bool warm = false, bool track_prev = false) {
  hipStream_t st = p->stream;
  p->lastB = B;
  {
    Timer tm(p, K_TUBE);
    TubeParams tp{B, p->n, p->m, p->N, p->pmax, p->ntheta, p->CKpow.p, p->Ttube.p, p->power.p, d_xbar0, d_e0, p->theta.p, p->prestatus.p};
    hipLaunchKernelGGL(tz_tube_kernel, dim3(B), dim3(64), 0, st, tp);
    if (p->tube_stack) { int rc = tube_stack_theta(p, B, d_e0, st); if (rc) return rc; }
    AffineParams ap{B, p->ntheta, p->nz, p->mi, p->npar, p->q.view(), p->h.view(), p->par.view(), p->par_lo.p, p->par_hi.p,
                    p->theta.p, p->qv.p, p->hv.p, p->prestatus.p};
    size_t total = (size_t)B * (p->nz + p->mi + p->npar);
    hipLaunchKernelGGL(tz_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ap);
  }
  {
    Timer tm(p, K_IPM);
    IpmParams ip = ipm_params(p, B, d_status, d_iters, warm, track_prev);
    // active-set readout (slack < multiplier) needs the complementarity products well below the slacks: three decades below the
    // stopping target, but never below 1e-6 tol (1e-16 at the default tolerance: what the loosest calibrated target used to give --
    // with the tighter targets of round 3 an unbounded 1e-3 asked degenerate problems for mu = 1e-18 and they ended TZ_NUMERICAL)
    if (d_active) { const double f = std::min(1.0, std::max(1e-3, 1e-6 * p->tol / ip.mu_tol)); ip.mu_tol *= f; ip.mu_floor *= f; }
    hipLaunchKernelGGL(p->ipm_fn, dim3(B), dim3(TZ_THREADS), p->lds_bytes, st, ip);
  }
  {
    Timer tm(p, K_FINISH);
    FinishParams fp{B, p->n, p->m, p->N, p->nz, p->mi, p->nzp, p->nc_rows, p->P.p, p->Dz.p, p->Phi.p, p->Gam.p,
                    p->r1.p, p->R2.p, p->r0, p->cost_scale, p->row_of.p, p->act_scale.p, d_xbar0, p->qv.p, p->x.p, p->s.p, p->lam.p,
                    d_status, d_v, d_xbar, d_cost, d_active, cost_stride, p->vpos.p, p->rec0.p, p->recx.p, p->recy.p};
    hipLaunchKernelGGL(tz_finish_kernel, dim3(B), dim3(64), 0, st, fp);
  }
  TZ_HIP(hipGetLastError());
  return TZ_OK;
}

int launch_plant(tz_problem* p, int B, const double* d_A, const double* d_Bm, const double* d_w, size_t w_stride,
                 const double* d_v, const double* d_xbar_pred, const int* d_status,
                 double* d_x, double* d_xbar, double* d_e, double* d_u, size_t u_stride, double* d_xout, size_t x_stride, int* d_sticky) {
  Timer tm(p, K_PLANT);
  PlantParams pp{B, p->n, p->m, p->N, p->K.p, d_A, d_Bm, d_v, d_xbar_pred, d_w, w_stride, d_status, d_x, d_xbar, d_e,
                 d_u, u_stride, d_xout, x_stride, d_sticky};
  hipLaunchKernelGGL(tz_plant_kernel, dim3((B + 63) / 64), dim3(64), 0, p->stream, pp);
  TZ_HIP(hipGetLastError());
  return TZ_OK;
}

}  // namespace

extern "C" {

int tz_abi_version(void) { return TZ_ABI_VERSION; }
const char* tz_last_error(void) { return g_err.c_str(); }

int tz_device_count(int* count) {
  if (!count) TZ_FAIL(TZ_ERR_INVALID, "count is null");
  TZ_HIP(hipGetDeviceCount(count));
  return TZ_OK;
}

int tz_identify_batch(int device, int32_t B, int32_t T, int32_t n, int32_t m, const double* u, const double* x,
                      const double* w_center, const double* K, int32_t k_shared,
                      double* C, double* s, double* sK, double* CK, int32_t* status, int mem) {
  if (!u || !x || !w_center || !C || !s || !status) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (B <= 0 || T < 2) TZ_FAIL(TZ_ERR_INVALID, "B must be positive and T at least 2");
  if (n < 1 || n > TZ_ID_NMAX || m < 1 || m > TZ_ID_MMAX) TZ_FAIL(TZ_ERR_UNSUPPORTED, "identification on the device (K0): dim_x must be 1..%d and dim_u 1..%d", TZ_ID_NMAX, TZ_ID_MMAX);
  if (K && (!sK || !CK)) TZ_FAIL(TZ_ERR_INVALID, "sK and CK are required when K is given");
  if (mem != TZ_MEM_HOST && mem != TZ_MEM_DEVICE) TZ_FAIL(TZ_ERR_INVALID, "mem must be TZ_MEM_HOST or TZ_MEM_DEVICE");
  int ndev = 0;
  TZ_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0) TZ_FAIL(TZ_ERR_HIP, "no HIP device visible: the TZDDPC hot path has no CPU fallback");
  if (device < 0 || device >= ndev) TZ_FAIL(TZ_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
  TZ_HIP(hipSetDevice(device));
  const int p = n + m;
  const size_t b = (size_t)B;
  IdentifyParams q{B, T, n, m, u, x, w_center, K, k_shared ? 1 : 0, C, s, K ? sK : nullptr, K ? CK : nullptr, status};
  DevBuf<double> du, dx, dw, dK, dC, ds, dsK, dCK; DevBuf<int> dst;
  if (mem == TZ_MEM_HOST) {
    TZ_HIP(du.upload(u, b * T * m)); TZ_HIP(dx.upload(x, b * T * n)); TZ_HIP(dw.upload(w_center, (size_t)n));
    if (K) TZ_HIP(dK.upload(K, (k_shared ? 1 : b) * m * n));
    TZ_HIP(dC.alloc(b * n * p)); TZ_HIP(ds.alloc(b * p)); TZ_HIP(dsK.alloc(b * n)); TZ_HIP(dCK.alloc(b * n * n)); TZ_HIP(dst.alloc(b));
    q.u = du.p; q.x = dx.p; q.wc = dw.p; q.K = K ? dK.p : nullptr; q.C = dC.p; q.s = ds.p; q.sK = K ? dsK.p : nullptr; q.CK = K ? dCK.p : nullptr; q.status = dst.p;
  }
  hipLaunchKernelGGL(tz_identify_kernel, dim3(B), dim3(64), 0, 0, q);
  TZ_HIP(hipGetLastError());
  if (mem == TZ_MEM_HOST) {
    TZ_HIP(hipMemcpy(C, dC.p, b * n * p * sizeof(double), hipMemcpyDeviceToHost));
    TZ_HIP(hipMemcpy(s, ds.p, b * p * sizeof(double), hipMemcpyDeviceToHost));
    if (K) { TZ_HIP(hipMemcpy(sK, dsK.p, b * n * sizeof(double), hipMemcpyDeviceToHost)); TZ_HIP(hipMemcpy(CK, dCK.p, b * n * n * sizeof(double), hipMemcpyDeviceToHost)); }
    TZ_HIP(hipMemcpy(status, dst.p, b * sizeof(int), hipMemcpyDeviceToHost));
  }
  return TZ_OK;
}

// ---- gain synthesis (tz_gain.hip.h): spectral radii of sampled closed loops, CCP ascent of ||A + B K||_F ------------------------
static int gain_batch(bool adversary, int device, int32_t S, int32_t n, int32_t ngen, const double* M0, const double* H,
                      const double* beta, int32_t max_iter, double* beta_out, double* val, int32_t* aux) {
  if (!M0 || !beta || !val || !aux || (ngen > 0 && !H) || (adversary && !beta_out)) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (S <= 0 || ngen < 0) TZ_FAIL(TZ_ERR_INVALID, "S must be positive, ngen non-negative");
  if (n < 1 || n > TZ_GN_NMAX) TZ_FAIL(TZ_ERR_UNSUPPORTED, "dim_x must be 1..%d", TZ_GN_NMAX);
  if (adversary && max_iter < 1) TZ_FAIL(TZ_ERR_INVALID, "max_iter must be positive");
  int ndev = 0;
  TZ_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0) TZ_FAIL(TZ_ERR_HIP, "no HIP device visible: the TZDDPC hot path has no CPU fallback");
  if (device < 0 || device >= ndev) TZ_FAIL(TZ_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
  TZ_HIP(hipSetDevice(device));
  const size_t s = (size_t)S, g = (size_t)ngen, n2 = (size_t)n * n;
  DevBuf<double> dM0, dH, dbeta, dbout, dval; DevBuf<int> daux;
  TZ_HIP(dM0.upload(M0, n2));
  if (g) { TZ_HIP(dH.upload(H, g * n2)); TZ_HIP(dbeta.upload(beta, s * g)); } else { TZ_HIP(dH.alloc(1)); TZ_HIP(dbeta.alloc(1)); }
  TZ_HIP(dval.alloc(s)); TZ_HIP(daux.alloc(s));
  if (adversary) TZ_HIP(dbout.alloc(std::max<size_t>(s * g, 1)));
  GainParams q{S, n, ngen, max_iter, dM0.p, dH.p, dbeta.p, adversary ? dbout.p : nullptr, dval.p, daux.p};
  const dim3 grid((S + 63) / 64), block(64);
  if (adversary) hipLaunchKernelGGL(tz_adversary_kernel, dim3(S), dim3(256), 0, 0, q);          // one workgroup per starting point, lane = generator
  else hipLaunchKernelGGL(tz_specrad_kernel, grid, block, 0, 0, q);
  TZ_HIP(hipGetLastError());
  TZ_HIP(hipMemcpy(val, dval.p, s * sizeof(double), hipMemcpyDeviceToHost));
  TZ_HIP(hipMemcpy(aux, daux.p, s * sizeof(int), hipMemcpyDeviceToHost));
  if (adversary && g) TZ_HIP(hipMemcpy(beta_out, dbout.p, s * g * sizeof(double), hipMemcpyDeviceToHost));
  return TZ_OK;
}

int tz_specrad_batch(int device, int32_t S, int32_t n, int32_t ngen, const double* M0, const double* H, const double* beta,
                     double* rho, int32_t* status) {
  return gain_batch(false, device, S, n, ngen, M0, H, beta, 0, nullptr, rho, status);
}

int tz_adversary_batch(int device, int32_t S, int32_t n, int32_t ngen, const double* M0, const double* H, const double* beta0,
                       int32_t max_iter, double* beta, double* fro, int32_t* steps) {
  return gain_batch(true, device, S, n, ngen, M0, H, beta0, max_iter, beta, fro, steps);
}

struct tz_genstack {
  int device = 0, n = 0, m = 0, N = 0, nseg = 0, rec = 0, nchunk = 0;
  int64_t G = 0;
  std::vector<int> seg_ptr;                 // literal order
  DevBuf<double> recs_sorted, recs_lit, recs_mf, recs_mfn, c0, cE, cZ, K, partial, in_e0, in_zeta, o_c, o_rx, o_ru, o_Z;
  DevBuf<int> src_lit, seg_chunk_ptr;
  DevBuf<GsChunk> chunks;
  DevBuf<GsChunkM> chunks_m;                // matrix-core layout (tz_genstack_mfma_kernel): groups of 4 generators, K rows appended
  bool mfma = false;
  int rows_mf = 0;                          // rows per generator in recs_mf: n + m (K rows appended) or n (formed in the kernel)
  size_t pcap = 0;                          // doubles allocated for `partial`
  bool have_cZ = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

int tz_genstack_create(int device, const tz_genstack_desc* d, tz_genstack** out) {
  if (!d || !out || !d->seg_ptr || !d->src || !d->m0 || !d->M || !d->c0 || !d->cE || !d->K) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (d->n < 1 || d->n > TZ_NMAX || d->m < 1 || d->m > TZ_MMAX) TZ_FAIL(TZ_ERR_UNSUPPORTED, "dim_x must be 1..%d and dim_u 1..%d", TZ_NMAX, TZ_MMAX);
  if (d->N < 1 || d->nseg < 1) TZ_FAIL(TZ_ERR_INVALID, "N and nseg must be positive");
  int ndev = 0;
  TZ_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0) TZ_FAIL(TZ_ERR_HIP, "no HIP device visible: the TZDDPC hot path has no CPU fallback");
  if (device < 0 || device >= ndev) TZ_FAIL(TZ_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
  TZ_HIP(hipSetDevice(device));
  std::unique_ptr<tz_genstack> g(new tz_genstack());
  const int n = d->n, m = d->m, p = n + m, rec = n * (1 + p);
  g->device = device; g->n = n; g->m = m; g->N = d->N; g->nseg = d->nseg; g->rec = rec;
  g->seg_ptr.assign(d->seg_ptr, d->seg_ptr + d->nseg + 1);
  const int64_t G = g->seg_ptr[d->nseg];
  g->G = G;
  for (int64_t i = 0; i < G; ++i) if (d->src[i] < -1 || d->src[i] > d->N) TZ_FAIL(TZ_ERR_INVALID, "src[%lld] out of range", (long long)i);
  // records [m0 | M]: literal order (tz_genstack_values) and sorted by (tube, source) + cut into chunks (tz_genstack_intervals)
  std::vector<double> lit((size_t)std::max<int64_t>(G, 1) * rec, 0.0), srt(lit.size(), 0.0);
  auto fill = [&](double* dst, int64_t gi) {
    for (int i = 0; i < n; ++i) { dst[i] = d->m0[(size_t)gi * n + i]; for (int c = 0; c < p; ++c) dst[n + i * p + c] = d->M[((size_t)gi * n + i) * p + c]; }
  };
  for (int64_t gi = 0; gi < G; ++gi) fill(&lit[(size_t)gi * rec], gi);
  std::vector<GsChunk> chunks; std::vector<int> scp((size_t)d->nseg + 1, 0);
  int64_t pos = 0;
  for (int k = 0; k < d->nseg; ++k) {
    std::vector<int64_t> idx;
    for (int64_t gi = g->seg_ptr[k]; gi < g->seg_ptr[k + 1]; ++gi) idx.push_back(gi);
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b2) { return d->src[a] < d->src[b2]; });
    size_t a = 0;
    while (a < idx.size()) {
      size_t b2 = a;
      while (b2 < idx.size() && d->src[idx[b2]] == d->src[idx[a]] && b2 - a < TZ_GS_CHUNK) ++b2;
      chunks.push_back(GsChunk{k, d->src[idx[a]], (int)pos, (int)(pos + (int64_t)(b2 - a))});
      for (size_t e = a; e < b2; ++e) fill(&srt[(size_t)pos++ * rec], idx[e]);
      a = b2;
    }
    scp[k + 1] = (int)chunks.size();
  }
  g->nchunk = (int)chunks.size();
  // matrix-core layout of the sorted stack (dimensions with a compiled instance): per group of 4 generators (a chunk is padded
  // with zero generators) [component c < P][generator i < 4][inner k < P] of Mext = [M; K M], then [c][i] of m0ext = [m0; K m0]
  g->mfma = (p >= 3 && p <= 7) && !tz_dev_getenv("TZ_GS_VALU");
  // one input (the reference's systems): a second copy of the stack holds only the n rows [m0 | M] -- 1 / (n + 1) fewer bytes and matrix
  // instructions -- and K g is formed in the kernel; it serves every batch except 33 .. 64 trajectories, where the extra vector
  // arithmetic of the narrow kernel costs more than the rows save (measured: tools/k1g_ab.sh).  TZ_GS_KROWS=1: appended rows only.
  g->rows_mf = (m == 1 && !tz_dev_getenv("TZ_GS_KROWS")) ? n : p;
  if (g->mfma && !chunks.empty()) {
    std::vector<GsChunkM> cm;
    std::vector<double> ext((size_t)p * (p + 1));
    auto layout = [&](int RW, std::vector<double>& mf, bool want_chunks) {
      const int GD = 4 * RW * (p + 1);
      for (const GsChunk& ch : chunks) {
        const int ng = ch.g1 - ch.g0, nq = (ng + 3) / 4;
        const size_t q0 = mf.size() / GD;
        mf.resize(mf.size() + (size_t)nq * GD, 0.0);
        for (int gi = 0; gi < ng; ++gi) {
          const double* r = &srt[(size_t)(ch.g0 + gi) * rec];            // [m0 (n) | M (n x p)]
          for (int c = 0; c < RW; ++c) {
            double m0e = 0.0;
            if (c < n) m0e = r[c]; else for (int i = 0; i < n; ++i) m0e += d->K[(size_t)(c - n) * n + i] * r[i];
            ext[(size_t)c * (p + 1)] = m0e;
            for (int k = 0; k < p; ++k) {
              double v = 0.0;
              if (c < n) v = r[n + c * p + k]; else for (int i = 0; i < n; ++i) v += d->K[(size_t)(c - n) * n + i] * r[n + i * p + k];
              ext[(size_t)c * (p + 1) + 1 + k] = v;
            }
          }
          double* gb = &mf[(q0 + gi / 4) * GD];
          const int i4 = gi & 3;
          for (int c = 0; c < RW; ++c) {
            for (int k = 0; k < p; ++k) gb[c * 4 * p + i4 * p + k] = ext[(size_t)c * (p + 1) + 1 + k];
            gb[4 * RW * p + 4 * c + i4] = ext[(size_t)c * (p + 1)];
          }
        }
        if (want_chunks) cm.push_back(GsChunkM{ch.seg, ch.src, (int)q0, nq});       // group offsets are the same in both layouts
      }
    };
    std::vector<double> mf;
    layout(p, mf, true);
    for (double v : mf) if (!std::isfinite(v)) TZ_FAIL(TZ_ERR_INVALID, "non-finite generator entry");
    TZ_HIP(g->recs_mf.upload(mf)); TZ_HIP(g->chunks_m.upload(cm));
    if (g->rows_mf != p) {
      std::vector<double> mfn;
      layout(g->rows_mf, mfn, false);
      TZ_HIP(g->recs_mfn.upload(mfn));
    }
  }
  TZ_HIP(g->recs_lit.upload(lit)); TZ_HIP(g->recs_sorted.upload(srt));
  TZ_HIP(g->src_lit.upload(d->src, (size_t)std::max<int64_t>(G, 1)));
  if (chunks.empty()) chunks.push_back(GsChunk{0, -1, 0, 0});
  TZ_HIP(g->chunks.upload(chunks)); TZ_HIP(g->seg_chunk_ptr.upload(scp));
  TZ_HIP(g->c0.upload(d->c0, (size_t)d->nseg * n)); TZ_HIP(g->cE.upload(d->cE, (size_t)d->nseg * n * n));
  if (d->cZ) {
    const size_t cnt = (size_t)d->nseg * d->N * n * p;
    for (size_t i = 0; i < cnt && !g->have_cZ; ++i) if (d->cZ[i] != 0.0) g->have_cZ = true;
    if (g->have_cZ) TZ_HIP(g->cZ.upload(d->cZ, cnt));
  }
  TZ_HIP(g->K.upload(d->K, (size_t)m * n));
  TZ_HIP(hipEventCreate(&g->ev0)); TZ_HIP(hipEventCreate(&g->ev1));
  *out = g.release();
  return TZ_OK;
}

int tz_genstack_destroy(tz_genstack* g) {
  if (!g) return TZ_OK;
  (void)hipSetDevice(g->device);
  (void)hipDeviceSynchronize();
  if (g->ev0) (void)hipEventDestroy(g->ev0);
  if (g->ev1) (void)hipEventDestroy(g->ev1);
  delete g;
  return TZ_OK;
}

int tz_genstack_info(tz_genstack* g, int64_t* generators, int64_t* stack_bytes, int64_t* chunks) {
  if (!g) TZ_FAIL(TZ_ERR_INVALID, "null handle");
  if (generators) *generators = g->G;
  if (stack_bytes) *stack_bytes = g->G * g->rec * (int64_t)sizeof(double);
  if (chunks) *chunks = g->nchunk;
  return TZ_OK;
}

namespace {
int gs_inputs(tz_genstack* g, int B, const double* e0, const double* zeta, int mem, const double** de0, const double** dz) {
  const size_t p = g->n + g->m;
  if (mem == TZ_MEM_DEVICE) { *de0 = e0; *dz = zeta; return TZ_OK; }
  if (mem != TZ_MEM_HOST) TZ_FAIL(TZ_ERR_INVALID, "mem must be TZ_MEM_HOST or TZ_MEM_DEVICE");
  TZ_HIP(g->in_e0.upload(e0, (size_t)B * g->n)); TZ_HIP(g->in_zeta.upload(zeta, (size_t)B * g->N * p));
  *de0 = g->in_e0.p; *dz = g->in_zeta.p;
  return TZ_OK;
}
}  // namespace

// evaluation of the whole stack for B trajectories, device pointers, on `st`
static int gs_eval(tz_genstack* g, int B, const double* de0, const double* dz, double* dc, double* drx, double* dru, hipStream_t st) {
  const int n = g->n, m = g->m, p = n + m;
  GenstackParams q{B, n, m, g->N, g->nseg, g->nchunk, g->rec, g->recs_sorted.p, g->chunks.p, g->K.p, de0, dz, g->partial.p};
  const dim3 grid((unsigned)g->nchunk, (unsigned)((B + 255) / 256));
  int nsub = 1;
  TZ_HIP(hipEventRecord(g->ev0, st));
  if (g->nchunk > 0 && g->mfma) {
    // few trajectories: every wave takes all of them and a quarter of the generators (the stack is streamed once: HBM-bound);
    // many: 256 per workgroup, the stack is re-read from L2 by the tiles of a chunk, which share an XCD
    const bool split = B <= 64;
    const int nq = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    nsub = split ? 2 : 1;                             // two blocks per chunk when every block streams its tiles once (measured at 32 trajectories,
    if (split) {                                      // 636 chunks: 1 -> 0.0670 ms, 2 -> 0.0641 ms, 4 -> 0.120 ms: the un-overlapped prologue of short blocks)
      const char* e = tz_dev_getenv("TZ_GS_NSUB");
      if (e) nsub = std::min(TZ_GS_MAXSUB, std::max(1, atoi(e)));
    }
    const int ntt = split ? nsub : (B + 255) / 256;
    const bool krows = g->rows_mf == p || (B > 32 && B <= 64);       // which copy of the stack: K rows appended, or formed in the kernel
    GenstackMParams qm{B, n, m, g->N, g->nchunk, ntt, nsub, krows ? g->recs_mf.p : g->recs_mfn.p, g->K.p, g->chunks_m.p, de0, dz, g->partial.p};
    const dim3 gm((unsigned)(((g->nchunk + 7) / 8) * 8 * ntt));
    const bool narrow = split && !tz_dev_getenv("TZ_GS_NO_NARROW");     // wave-private pipeline (no barrier in the stream); the switch keeps the barrier form
#define TZ_GS_LAUNCH(RR, PP) do { \
      if (!split) hipLaunchKernelGGL((tz_genstack_mfma_kernel<RR, PP, 4, false>), gm, dim3(256), 0, st, qm); \
      else if (narrow && nq == 1) hipLaunchKernelGGL((tz_genstack_mfma_narrow_kernel<RR, PP, 1>), gm, dim3(256), 0, st, qm); \
      else if (narrow && nq == 2) hipLaunchKernelGGL((tz_genstack_mfma_narrow_kernel<RR, PP, 2>), gm, dim3(256), 0, st, qm); \
      else if (narrow) hipLaunchKernelGGL((tz_genstack_mfma_narrow_kernel<RR, PP, 4>), gm, dim3(256), 0, st, qm); \
      else if (nq == 1) hipLaunchKernelGGL((tz_genstack_mfma_kernel<RR, PP, 1, true>), gm, dim3(256), 0, st, qm); \
      else if (nq == 2) hipLaunchKernelGGL((tz_genstack_mfma_kernel<RR, PP, 2, true>), gm, dim3(256), 0, st, qm); \
      else hipLaunchKernelGGL((tz_genstack_mfma_kernel<RR, PP, 4, true>), gm, dim3(256), 0, st, qm); } while (0)
    if (krows) {
      switch (p) {
        case 3: TZ_GS_LAUNCH(3, 3); break;
        case 4: TZ_GS_LAUNCH(4, 4); break;
        case 5: TZ_GS_LAUNCH(5, 5); break;
        case 6: TZ_GS_LAUNCH(6, 6); break;
        default: TZ_GS_LAUNCH(7, 7); break;
      }
    } else {
      switch (p) {
        case 3: TZ_GS_LAUNCH(2, 3); break;
        case 4: TZ_GS_LAUNCH(3, 4); break;
        case 5: TZ_GS_LAUNCH(4, 5); break;
        case 6: TZ_GS_LAUNCH(5, 6); break;
        default: TZ_GS_LAUNCH(6, 7); break;
      }
    }
#undef TZ_GS_LAUNCH
  } else if (g->nchunk > 0) {
    if (n == 2 && m == 1) hipLaunchKernelGGL((tz_genstack_kernel<2, 1>), grid, dim3(256), 0, st, q);
    else if (n == 4 && m == 1) hipLaunchKernelGGL((tz_genstack_kernel<4, 1>), grid, dim3(256), 0, st, q);
    else if (n == 5 && m == 1) hipLaunchKernelGGL((tz_genstack_kernel<5, 1>), grid, dim3(256), 0, st, q);
    else hipLaunchKernelGGL((tz_genstack_kernel<0, 0>), grid, dim3(256), 0, st, q);
  }
  TZ_HIP(hipEventRecord(g->ev1, st));          // the stream kernel alone (what rocprofv3 reports for it); the reduction follows
  GsReduceParams r{B, n, m, g->N, g->nseg, nsub, g->seg_chunk_ptr.p, g->partial.p, g->c0.p, g->cE.p, g->have_cZ ? g->cZ.p : nullptr, de0, dz, dc, drx, dru};
  const size_t total = (size_t)B * g->nseg * p;
  if (B <= 64 && n <= 16) hipLaunchKernelGGL(tz_genstack_reduce16_kernel, dim3((unsigned)((total * 16 + 255) / 256)), dim3(256), 0, st, r);   // few trajectories: latency, not bytes
  else hipLaunchKernelGGL(tz_genstack_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r);
  TZ_HIP(hipGetLastError());
  return TZ_OK;
}

int tz_genstack_intervals(tz_genstack* g, int32_t B, const double* e0, const double* zeta,
                          double* centre, double* rad_x, double* rad_u, double* kernel_ms, int mem) {
  if (!g || !e0 || !zeta || !centre || !rad_x || !rad_u) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (B <= 0) TZ_FAIL(TZ_ERR_INVALID, "batch size must be positive");
  TZ_HIP(hipSetDevice(g->device));
  const int n = g->n, m = g->m, p = n + m;
  const double *de0 = nullptr, *dz = nullptr;
  int rc = gs_inputs(g, B, e0, zeta, mem, &de0, &dz);
  if (rc) return rc;
  { const size_t need = (size_t)std::max(g->nchunk, 1) * (B <= 64 ? TZ_GS_MAXSUB : 1) * B * p; if (need > g->pcap) { TZ_HIP(g->partial.alloc(need)); g->pcap = need; } }
  double *dc = centre, *drx = rad_x, *dru = rad_u;
  if (mem == TZ_MEM_HOST) {
    TZ_HIP(g->o_c.alloc((size_t)B * g->nseg * n)); TZ_HIP(g->o_rx.alloc((size_t)B * g->nseg * n)); TZ_HIP(g->o_ru.alloc((size_t)B * g->nseg * m));
    dc = g->o_c.p; drx = g->o_rx.p; dru = g->o_ru.p;
  }
  int rc2 = gs_eval(g, B, de0, dz, dc, drx, dru, 0);
  if (rc2) return rc2;
  if (mem == TZ_MEM_HOST) {
    TZ_HIP(hipMemcpy(centre, dc, (size_t)B * g->nseg * n * sizeof(double), hipMemcpyDeviceToHost));
    TZ_HIP(hipMemcpy(rad_x, drx, (size_t)B * g->nseg * n * sizeof(double), hipMemcpyDeviceToHost));
    TZ_HIP(hipMemcpy(rad_u, dru, (size_t)B * g->nseg * m * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (kernel_ms) {
    TZ_HIP(hipEventSynchronize(g->ev1));
    float ms = 0.f;
    TZ_HIP(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    *kernel_ms = ms;
  }
  return TZ_OK;
}

int tube_stack_theta(tz_problem* p, int B, const double* d_e0, hipStream_t st) {
  tz_genstack* g = p->tube_stack;
  const int n = p->n, m = p->m, pq = n + m;
  if (B > p->ts_cap) {
    TZ_HIP(p->ts_zeta.alloc((size_t)B * g->N * pq)); TZ_HIP(hipMemset(p->ts_zeta.p, 0, (size_t)B * g->N * pq * sizeof(double)));
    TZ_HIP(p->ts_c.alloc((size_t)B * g->nseg * n)); TZ_HIP(p->ts_rx.alloc((size_t)B * g->nseg * n)); TZ_HIP(p->ts_ru.alloc((size_t)B * g->nseg * m));
    p->ts_cap = B;
  }
  { const size_t need = (size_t)std::max(g->nchunk, 1) * (B <= 64 ? TZ_GS_MAXSUB : 1) * B * pq; if (need > g->pcap) { TZ_HIP(g->partial.alloc(need)); g->pcap = need; } }
  int rc = gs_eval(g, B, d_e0, p->ts_zeta.p, p->ts_c.p, p->ts_rx.p, p->ts_ru.p, st);
  if (rc) return rc;
  ThetaStackParams q{B, n, m, p->N, g->nseg, p->ntheta, p->ts_c.p, p->ts_rx.p, p->ts_ru.p, p->theta.p};
  const size_t total = (size_t)B * p->N * (2 * n + m);
  hipLaunchKernelGGL(tz_theta_stack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, q);
  TZ_HIP(hipGetLastError());
  return TZ_OK;
}

int tz_problem_attach_tube_stack(tz_problem* p, tz_genstack* g) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  if (g) {
    if (g->device != p->device || g->n != p->n || g->m != p->m) TZ_FAIL(TZ_ERR_INVALID, "stack and problem differ in device or dimensions");
    if (g->nseg < p->N) TZ_FAIL(TZ_ERR_INVALID, "the stack holds %d tubes, the problem needs %d", g->nseg, p->N);
    p->fuse_enabled = false;                                   // the fused step computes theta in-kernel from the collapsed recursion
  }
  p->tube_stack = g;
  return TZ_OK;
}

int tz_genstack_values(tz_genstack* g, int32_t seg, int32_t B, const double* e0, const double* zeta, double* Z, int mem) {
  if (!g || !e0 || !zeta || !Z) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (B <= 0 || seg < 0 || seg >= g->nseg) TZ_FAIL(TZ_ERR_INVALID, "bad batch size or tube index");
  TZ_HIP(hipSetDevice(g->device));
  const int n = g->n, m = g->m, p = n + m;
  const double *de0 = nullptr, *dz = nullptr;
  int rc = gs_inputs(g, B, e0, zeta, mem, &de0, &dz);
  if (rc) return rc;
  const int ngen = g->seg_ptr[seg + 1] - g->seg_ptr[seg];
  const size_t cnt = (size_t)B * n * (1 + ngen);
  double* dZ = Z;
  if (mem == TZ_MEM_HOST) { TZ_HIP(g->o_Z.alloc(cnt)); dZ = g->o_Z.p; }
  GsValuesParams q{B, n, m, g->N, seg, ngen, g->rec, g->recs_lit.p + (size_t)g->seg_ptr[seg] * g->rec, g->src_lit.p + g->seg_ptr[seg],
                   g->c0.p + (size_t)seg * n, g->cE.p + (size_t)seg * n * n, g->have_cZ ? g->cZ.p + (size_t)seg * g->N * n * p : nullptr, de0, dz, dZ};
  const size_t total = (size_t)B * (1 + ngen);
  hipLaunchKernelGGL(tz_genstack_values_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, q);
  TZ_HIP(hipGetLastError());
  if (mem == TZ_MEM_HOST) TZ_HIP(hipMemcpy(Z, dZ, cnt * sizeof(double), hipMemcpyDeviceToHost));
  return TZ_OK;
}

int tz_problem_create(int device, const tz_problem_desc* d, tz_problem** out) {
  if (!d || !out) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (d->abi_version != TZ_ABI_VERSION) TZ_FAIL(TZ_ERR_INVALID, "abi_version %d != %d", d->abi_version, TZ_ABI_VERSION);
  if (d->pmax > TZ_PMAX) TZ_FAIL(TZ_ERR_UNSUPPORTED, "pmax=%d > %d powers of M_K not supported by tz_tube_kernel", d->pmax, TZ_PMAX);
  if (d->n < 1 || d->n > TZ_NMAX || d->m < 1 || d->m > TZ_MMAX) TZ_FAIL(TZ_ERR_UNSUPPORTED, "dim_x must be 1..%d and dim_u 1..%d", TZ_NMAX, TZ_MMAX);
  if (d->N < 1 || d->nz < d->N * d->m || d->mi < 1) TZ_FAIL(TZ_ERR_INVALID, "inconsistent sizes N=%d nz=%d mi=%d", d->N, d->nz, d->mi);
  if (d->nz > 256) TZ_FAIL(TZ_ERR_UNSUPPORTED, "nz=%d > 256 decision variables not supported by tz_ipm_kernel", d->nz);
  if (d->mi > 6 * TZ_THREADS) TZ_FAIL(TZ_ERR_UNSUPPORTED, "mi=%d > %d inequality rows not supported by tz_ipm_kernel", d->mi, 6 * TZ_THREADS);
  if (d->ntheta != 2 * d->n + d->N * (2 * d->n + d->m)) TZ_FAIL(TZ_ERR_INVALID, "ntheta mismatch");
  if (d->q.rows != d->nz || d->h.rows != d->mi) TZ_FAIL(TZ_ERR_INVALID, "affine map row counts do not match nz / mi");
  for (int k = 0; k < d->N; ++k)
    if (d->power[k] < 0 || d->power[k] > d->pmax) TZ_FAIL(TZ_ERR_INVALID, "power[%d]=%d outside 0..pmax", k, d->power[k]);
  for (int r = 0; r < d->mi; ++r)
    if (d->row_of[r] < 0 || d->row_of[r] >= std::max(d->nc_rows, 1)) TZ_FAIL(TZ_ERR_INVALID, "row_of[%d] out of range", r);
  int ndev = 0;
  TZ_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0) TZ_FAIL(TZ_ERR_HIP, "no HIP device visible: the TZDDPC hot path has no CPU fallback");
  if (device < 0 || device >= ndev) TZ_FAIL(TZ_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
  TZ_HIP(hipSetDevice(device));

  tz_problem* p = new tz_problem();
  std::unique_ptr<tz_problem> guard(p);
  p->device = device;
  p->n = d->n; p->m = d->m; p->N = d->N; p->nz = d->nz; p->mi = d->mi; p->ntheta = d->ntheta;
  p->npar = d->par.rows; p->nc_rows = d->nc_rows; p->pmax = d->pmax;
  p->max_iter = d->max_iter > 0 ? d->max_iter : 40;
  p->tol = d->tol > 0 ? d->tol : 1e-10; p->reg = d->reg > 0 ? d->reg : 1e-12;
  p->step_frac = (d->step_frac > 0 && d->step_frac < 1) ? d->step_frac : 0.99;
  p->cost_scale = d->cost_scale; p->r0 = d->r0;
  const int nz = d->nz, mi = d->mi;
  p->Tz = (nz + 3) / 4; p->nzp = 4 * p->Tz;
  p->Kc = (mi + 3) / 4; p->mip = 4 * p->Kc;
  const int Tz = p->Tz, Kc = p->Kc, nzp = p->nzp, mip = p->mip;
  p->nquads = 0;
  for (int I = 0; I < Tz; ++I) p->nquads += (I >> 2) + 1;

  p->maxr = (mi + TZ_THREADS - 1) / TZ_THREADS; p->ncg = (nzp + 63) / 64;
  p->tt = (p->ncg >= 2 || p->maxr > 4);
  // Staircase ordering (tile-triangle class): the library keeps the variables in time order (v_k next to the epigraph variables
  // of step k) and the rows by their last non-zero column, so that the non-zeros of G lie under a staircase: super-step s (16
  // rows) touches only the tile columns 0 .. cmax[s], non-decreasing in s.  A unit of the blocked Gram is then active on a
  // contiguous range of super-steps [s0, S) and runs there without a single mask test.  Purely structural: the time of v[k, j]
  // (the first N m variables, reference tzddpc/tzddpc.py:155) is k, the time of any other variable the smallest, over the rows
  // it appears in, of the latest input in that row.  Callers never see the ordering (outputs go through vpos / row_of).
  std::vector<int> permc((size_t)nz), permr((size_t)mi), invc((size_t)nz), invr((size_t)mi);
  std::iota(permc.begin(), permc.end(), 0); std::iota(permr.begin(), permr.end(), 0);
  const int nv = d->N * d->m;
  if (p->tt && !(d->plan_flags & TZ_PLAN_NO_STAIRCASE) && tz_dev_getenv("TZ_NO_STAIRCASE") == nullptr) {
    std::vector<int> rowt((size_t)mi, -1), colt((size_t)nz, 1 << 30);
    for (int r = 0; r < mi; ++r) for (int c = 0; c < nv; ++c) if (d->G[(size_t)r * nz + c] != 0.0) rowt[r] = std::max(rowt[r], c / d->m);
    for (int c = 0; c < nv; ++c) colt[c] = c / d->m;
    for (int c = nv; c < nz; ++c) { for (int r = 0; r < mi; ++r) if (d->G[(size_t)r * nz + c] != 0.0) colt[c] = std::min(colt[c], rowt[r]); if (colt[c] == (1 << 30)) colt[c] = d->N; }
    std::stable_sort(permc.begin(), permc.end(), [&](int a, int b) { return colt[a] < colt[b]; });
    for (int i = 0; i < nz; ++i) invc[permc[i]] = i;
    std::vector<int> last((size_t)mi, -1);
    for (int r = 0; r < mi; ++r) for (int c = 0; c < nz; ++c) if (d->G[(size_t)r * nz + c] != 0.0) last[r] = std::max(last[r], invc[c]);
    std::stable_sort(permr.begin(), permr.end(), [&](int a, int b) { return last[a] < last[b]; });
    p->permc = permc; p->permr = permr; p->staircase = true;
    std::vector<int> vp((size_t)nv);
    for (int c = 0; c < nv; ++c) vp[c] = invc[c];
    TZ_HIP(p->vpos.upload(vp));
  }
  for (int i = 0; i < nz; ++i) invc[permc[i]] = i;
  for (int i = 0; i < mi; ++i) invr[permr[i]] = i;
  // padded dense copies (device order)
  std::vector<double> P((size_t)nzp * nzp, 0.0), G((size_t)mip * nzp, 0.0), Gp((size_t)(Kc + 1) * (Tz + 1) * 16, 0.0);   // tile Tz of every row and the last patch row stay zero (masked operands / prefetch padding)
  for (int r = 0; r < nz; ++r) for (int c = 0; c < nz; ++c) P[(size_t)r * nzp + c] = d->P[(size_t)permc[r] * nz + permc[c]];
  for (int r = 0; r < mi; ++r) for (int c = 0; c < nz; ++c) {
    double v = d->G[(size_t)permr[r] * nz + permc[c]];
    G[(size_t)r * nzp + c] = v;
    Gp[((size_t)(r >> 2) * (Tz + 1) + (c >> 2)) * 16 + 4 * (r & 3) + (c & 3)] = v;
  }
  p->nP = 0;
  for (int r = 0; r < nz; ++r) for (int c = 0; c < nz; ++c) if (P[(size_t)r * nzp + c] != 0.0) p->nP = r + 1;
  // Gram plan: item = (block of 4 tile rows I0..I0+3, quads q0..q0+nq-1), k-list = chunks where the 16 columns are non-zero
  std::vector<int> klist;
  std::vector<IpmItem> items;
  std::vector<double> cost;
  const int NB = (Tz + 3) / 4;
  for (int IB = 0; IB < NB; ++IB) {
    const int kptr = (int)klist.size();
    for (int kc = 0; kc < Kc; ++kc) {
      bool nzr = false;
      for (int r = 4 * kc; r < std::min(4 * kc + 4, mi) && !nzr; ++r)
        for (int c = 16 * IB; c < std::min(16 * IB + 16, nz); ++c)
          if (G[(size_t)r * nzp + c] != 0.0) { nzr = true; break; }
      if (nzr) klist.push_back(kc);
    }
    const int klen = (int)klist.size() - kptr;
    for (int z = 0; z < 8; ++z) klist.push_back(Kc);     // prefetch padding: the all-zero patch row
    const int Ilast = std::min(4 * IB + 3, Tz - 1);
    const int qmax = Ilast >> 2;          // quads 0..qmax exist for the last row of the block
    for (int q0 = 0; q0 <= qmax; q0 += 2) {
      IpmItem it{4 * IB, q0, std::min(2, qmax - q0 + 1), kptr, klen};
      items.push_back(it);
      cost.push_back((double)klen * 4 * it.nq + 8);
    }
  }
  // LPT assignment to the 4 waves
  std::vector<int> order(items.size());
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  std::vector<std::vector<int>> per_wave(TZ_NWAVES);
  double load[TZ_NWAVES] = {0, 0, 0, 0};
  for (int idx : order) {
    int w = (int)(std::min_element(load, load + TZ_NWAVES) - load);
    per_wave[w].push_back(idx); load[w] += cost[idx];
  }
  std::vector<IpmItem> items_sorted;
  std::vector<int> item_ptr(TZ_NWAVES + 1, 0);
  p->mfma_gram = 0;
  for (int w = 0; w < TZ_NWAVES; ++w) {
    for (int idx : per_wave[w]) {
      const IpmItem& it = items[idx];
      items_sorted.push_back(it);
      const int validI = std::min(4, Tz - it.I0);
      p->mfma_gram += (int64_t)it.klen * validI * it.nq;
      p->mfma_issued += (int64_t)it.klen * 8;
    }
    item_ptr[w + 1] = (int)items_sorted.size();
  }
  p->mfma_chol = 0;
  for (int pp = 0; pp < Tz; ++pp)
    for (int I = pp + 1; I < Tz; ++I) p->mfma_chol += (I >> 2) - ((pp + 1) >> 2) + 1;
  p->mfma_issued += p->mfma_chol;

  TZ_HIP(p->P.upload(P)); TZ_HIP(p->Gp.upload(Gp));
  {
    const int S = (Kc + 3) / 4;
    std::vector<int> sm((size_t)S + 1, 0);
    if (Tz <= 31) for (int r = 0; r < mi; ++r) for (int c = 0; c < nz; ++c) if (G[(size_t)r * nzp + c] != 0.0) sm[r >> 4] |= 1 << (c >> 2);
    TZ_HIP(p->smask.upload(sm));
  }
  {
    std::vector<std::vector<std::pair<int, double>>> byrow((size_t)mi), bycol((size_t)nz);
    for (int r = 0; r < mi; ++r) for (int c = 0; c < nz; ++c) {
      const double v = G[(size_t)r * nzp + c];
      if (v != 0.0) { byrow[r].push_back({c, v}); bycol[c].push_back({r, v}); }
    }
    // G x: twice as many virtual lanes as rows, so the long rows can be cut up;  G'v: the 192 lanes of waves 1-3 per pass
    TZ_HIP(p->eg.build(byrow, TZ_THREADS, 2 * mi, p->tt));
    TZ_HIP(p->et.build(bycol, TZ_THREADS - 64, std::max(TZ_THREADS - 64, 2 * nz), p->tt));
    p->nell = std::max(p->eg.VL, p->et.VL);
  }
  if (klist.empty()) klist.push_back(0);
  p->nklist = std::max((int)klist.size(), (Kc + 3) / 4 + 1);     // the LDS k-list area doubles as the super-step mask table (ksplit)
  TZ_HIP(p->klist.upload(klist)); TZ_HIP(p->items.upload(items_sorted)); TZ_HIP(p->item_ptr.upload(item_ptr));
  TZ_HIP(p->q.upload(d->q, permc.data())); TZ_HIP(p->h.upload(d->h, permr.data())); TZ_HIP(p->par.upload(d->par));
  TZ_HIP(p->par_lo.upload(d->par_lo, (size_t)p->npar)); TZ_HIP(p->par_hi.upload(d->par_hi, (size_t)p->npar));
  TZ_HIP(p->Dz.upload(d->Dz, (size_t)nz));
  if (d->rec_y || d->rec_c0 || d->rec_x0) {                              // equality-eliminated problem: affine recovery of v
    if (!(d->rec_y && d->rec_c0 && d->rec_x0)) TZ_FAIL(TZ_ERR_INVALID, "rec_c0, rec_x0 and rec_y go together");
    std::vector<double> ry((size_t)nv * nz);
    for (int c = 0; c < nv; ++c) for (int k = 0; k < nz; ++k) ry[(size_t)c * nz + k] = d->rec_y[(size_t)c * nz + permc[k]];
    TZ_HIP(p->recy.upload(ry)); TZ_HIP(p->rec0.upload(d->rec_c0, (size_t)nv)); TZ_HIP(p->recx.upload(d->rec_x0, (size_t)nv * d->n));
  }
  {                                                                      // xbar[1] = Phi_1 xbar0 + Gam_1 v with Gam_1 confined to v[0] (it is A xbar0 + B v[0], reference :166-170)
    bool only_v0 = !(d->rec_y || d->rec_c0 || d->rec_x0);
    for (int i = 0; i < d->n && only_v0; ++i) for (int c = d->m; c < nv; ++c) if (d->Gam[((size_t)d->n + i) * nv + c] != 0.0) { only_v0 = false; break; }
    p->lean_epilogue = only_v0;
  }
  TZ_HIP(p->Phi.upload(d->Phi, (size_t)(d->N + 1) * d->n * d->n));
  TZ_HIP(p->Gam.upload(d->Gam, (size_t)(d->N + 1) * d->n * d->N * d->m));
  TZ_HIP(p->r1.upload(d->r1, (size_t)d->n)); TZ_HIP(p->R2.upload(d->R2, (size_t)d->n * d->n));
  TZ_HIP(p->CK.upload(d->CK, (size_t)d->n * d->n)); TZ_HIP(p->DK.upload(d->DK, (size_t)d->n * d->n));
  TZ_HIP(p->K.upload(d->K, (size_t)d->m * d->n));
  {
    // Resolvent of the tube recursion (tz_kernels.hip.h, TubeParams): with X_j = |C_K^j|, U_j = |K C_K^j|,
    //   R_0 = D_K, R_d = D_K Tx_{d-1};  Tx_d = sum_{j<=d} X_{d-j} R_j;  Tu_d = sum_{j<=d} U_{d-j} R_j;  and C_K^l.
    const int n = d->n, m = d->m, pm = d->pmax, nm = n + m;
    std::vector<double> ckp((size_t)(pm + 1) * n * n, 0.0), T((size_t)std::max(pm, 1) * nm * n, 0.0), R((size_t)std::max(pm, 1) * n * n, 0.0);
    for (int i = 0; i < n; ++i) ckp[(size_t)i * n + i] = 1.0;
    for (int l = 1; l <= pm; ++l)
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        double a = 0.0;
        for (int k = 0; k < n; ++k) a += d->CK[i * n + k] * ckp[((size_t)(l - 1) * n + k) * n + j];
        ckp[((size_t)l * n + i) * n + j] = a;
      }
    for (int dd = 0; dd < pm; ++dd) {
      double* Rd = &R[(size_t)dd * n * n];
      if (dd == 0) for (int e = 0; e < n * n; ++e) Rd[e] = d->DK[e];
      else {
        const double* Tp = &T[(size_t)(dd - 1) * nm * n];      // Tx_{d-1} = first n rows
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
          double a = 0.0;
          for (int k = 0; k < n; ++k) a += d->DK[i * n + k] * Tp[k * n + j];
          Rd[i * n + j] = a;
        }
      }
      double* Td = &T[(size_t)dd * nm * n];
      for (int jj = 0; jj <= dd; ++jj) {
        const double* X = d->absCKpow + (size_t)(dd - jj) * n * n;
        const double* U = d->absKCKpow + (size_t)(dd - jj) * m * n;
        const double* Rj = &R[(size_t)jj * n * n];
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
          double a = 0.0;
          for (int k = 0; k < n; ++k) a += X[i * n + k] * Rj[k * n + j];
          Td[i * n + j] += a;
        }
        for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
          double a = 0.0;
          for (int k = 0; k < n; ++k) a += U[i * n + k] * Rj[k * n + j];
          Td[(n + i) * n + j] += a;
        }
      }
    }
    TZ_HIP(p->CKpow.upload(ckp.data(), ckp.size()));
    TZ_HIP(p->Ttube.upload(T.data(), T.size()));
  }
  TZ_HIP(p->power.upload(d->power, (size_t)d->N));
  {
    std::vector<int> ro((size_t)mi); std::vector<double> as((size_t)mi);
    for (int i = 0; i < mi; ++i) { ro[i] = d->row_of[permr[i]]; as[i] = d->act_scale[permr[i]]; }
    TZ_HIP(p->row_of.upload(ro)); TZ_HIP(p->act_scale.upload(as));
  }
  if (d->shift_var && d->shift_row && d->shift_xscale && d->shift_lscale) {
    for (int c = 0; c < nz; ++c) if (d->shift_var[c] < 0 || d->shift_var[c] >= nz) TZ_FAIL(TZ_ERR_INVALID, "shift_var[%d] out of range", c);
    for (int r = 0; r < mi; ++r) if (d->shift_row[r] < 0 || d->shift_row[r] >= mi) TZ_FAIL(TZ_ERR_INVALID, "shift_row[%d] out of range", r);
    std::vector<int> sv((size_t)nz), sr((size_t)mi); std::vector<double> xs((size_t)nz), ls((size_t)mi);
    for (int i = 0; i < nz; ++i) { sv[i] = invc[d->shift_var[permc[i]]]; xs[i] = d->shift_xscale[permc[i]]; }
    for (int i = 0; i < mi; ++i) { sr[i] = invr[d->shift_row[permr[i]]]; ls[i] = d->shift_lscale[permr[i]]; }
    TZ_HIP(p->shift_var.upload(sv)); TZ_HIP(p->shift_row.upload(sr));
    TZ_HIP(p->shift_xs.upload(xs)); TZ_HIP(p->shift_ls.upload(ls));
    p->have_shift = true;
  }

  p->ntube = (d->pmax + 1) * d->n * d->n + std::max(d->pmax, 1) * (d->n + d->m) * d->n         // tube tables
             + 3 * d->n * d->n + 2 * d->n * d->m + d->n + d->n * d->N * d->m + d->N * d->m        // recovery / plant constants
             + (d->N + 1) / 2;                                                                    // power[k] (ints)
  const size_t LDS_MAX = 160 * 1024;
  auto fail_lds = [&](size_t need) { char b[256]; snprintf(b, sizeof(b), "not supported: the problem needs %zu bytes of LDS per workgroup (nz=%d, mi=%d); limit is 160 KiB", need, nz, mi); g_err = b; return TZ_ERR_UNSUPPORTED; };
  int wgs_per_cu = 1;
  if (p->tt) {
    // ---- tile-triangle class: masks and work plan of the blocked Gram, tile stride, workgroups per CU ----------------------
    p->ksplit = false; p->chol1 = false;
    p->ntile = Tz * (Tz + 1) / 2;
    const int S = (Kc + 3) / 4;
    // staircase: last non-zero tile column of every super-step, made non-decreasing (it is, when the ordering above is on);
    // sfirst[I] = first super-step that touches tile column I
    std::vector<int> cmaxs((size_t)S, -1);
    for (int r = 0; r < mi; ++r) for (int c = 0; c < nz; ++c) if (G[(size_t)r * nzp + c] != 0.0) cmaxs[r >> 4] = std::max(cmaxs[r >> 4], c >> 2);
    std::vector<int> sfirst((size_t)Tz + 1, S);
    for (int sIdx = S - 1; sIdx >= 0; --sIdx) for (int I = 0; I <= cmaxs[sIdx]; ++I) sfirst[I] = sIdx;
    // units: the tile index range is cut into near-equal ranges of at most TZ_GU tiles; a unit is a pair of ranges (row range >=
    // column range).  Cost = MFMAs it issues (masked); LPT over the waves; the range size with the smallest makespan wins.
    // units: the tile index range (padded with all-zero tile columns up to a multiple of U) is cut into ranges of exactly U tiles;
    // a unit is a pair of ranges (row range >= column range).  Cost = MFMAs + loads it issues; LPT over the waves; the U in
    // [UMAX - 2, UMAX] with the smallest makespan wins.
    std::vector<TzGUnit> best_units; std::vector<int> best_ptr; double best_span = 1e300; int best_u = 0;
    auto plan = [&](int UMAX) {
    best_units.clear(); best_span = 1e300;
    for (int U = std::max(1, UMAX - 2); U <= UMAX; ++U) {
      const int nrange = (Tz + U - 1) / U;
      std::vector<TzGUnit> units; std::vector<double> cost;
      for (int a = 0; a < nrange; ++a) for (int b = 0; b <= a; ++b) {
        TzGUnit u{a * U, b * U, sfirst[a * U]};
        const double per = (a == b) ? 0.5 * U * (U + 1) + 0.4 * U : (double)U * U + 0.4 * 2 * U;   // MFMAs + loads of one super-step
        units.push_back(u); cost.push_back(per * (S - u.s0 + 3) + 60.0);   // + pipeline fill, fold / store
      }
      std::vector<int> order(units.size());
      std::iota(order.begin(), order.end(), 0);
      std::sort(order.begin(), order.end(), [&](int x, int y) { return cost[x] > cost[y]; });
      std::vector<std::vector<int>> per_wave(TZ_NWAVES); double load[TZ_NWAVES] = {0, 0, 0, 0};
      for (int idx : order) { int w = (int)(std::min_element(load, load + TZ_NWAVES) - load); per_wave[w].push_back(idx); load[w] += cost[idx]; }
      const double span = *std::max_element(load, load + TZ_NWAVES);
      if (span < best_span) {
        best_span = span; best_u = U; best_units.clear(); best_ptr.assign(TZ_NWAVES + 1, 0);
        for (int w = 0; w < TZ_NWAVES; ++w) { for (int idx : per_wave[w]) best_units.push_back(units[idx]); best_ptr[w + 1] = (int)best_units.size(); }
      }
    }
    };
    p->nklist = 2;
    // LDS: two workgroups per CU with the padded tile stride if that fits, else one; the partial-sum buffer of the G x product
    // shrinks from two virtual lanes per row to one before the tile stride loses its padding
    auto lds_for = [&](int TS, int nell) { return tz_ipm_lds_doubles((size_t)p->ntile * TS, 1, Tz, nzp, mip, p->nklist, p->ntheta, 0, p->ntube, nell) * sizeof(double); };
    const int nell_full = p->nell;
    DevEll eg_small; int nell_small = nell_full;
    bool small_built = false;
    auto need_small = [&]() -> hipError_t {
      if (small_built) return hipSuccess;
      std::vector<std::vector<std::pair<int, double>>> byrow((size_t)mi);
      for (int r = 0; r < mi; ++r) for (int c = 0; c < nz; ++c) { const double v = G[(size_t)r * nzp + c]; if (v != 0.0) byrow[r].push_back({c, v}); }
      hipError_t e = eg_small.build(byrow, TZ_THREADS, mi, p->tt);
      nell_small = std::max(eg_small.VL, p->et.VL); small_built = true;
      return e;
    };
    struct Cand { int TS; bool small; int wgs; };
    const Cand cands[] = {{17, false, 2}, {17, true, 2}, {17, false, 1}, {17, true, 1}, {16, true, 1}};
    bool placed = false;
    for (const Cand& c : cands) {
      if (c.small) TZ_HIP(need_small());
      const size_t need = lds_for(c.TS, c.small ? nell_small : nell_full);
      if (need * c.wgs <= LDS_MAX) {
        p->TS = c.TS; wgs_per_cu = c.wgs; p->lds_bytes = need;
        if (c.small) { p->eg.swap(eg_small); p->nell = nell_small; }
        placed = true; break;
      }
    }
    if (!placed) { TZ_HIP(need_small()); return fail_lds(lds_for(16, nell_small)); }
    p->hsize = (size_t)p->ntile * p->TS;
    bool ttk = false;
    p->ipm_fn = ipm_kernel_for(p->maxr, p->ncg, wgs_per_cu, ttk);
    if (p->maxr > 4) wgs_per_cu = 1;                                   // those variants exist for one workgroup per CU only
    plan(TZ_TT_GU(wgs_per_cu));                                        // unit size <= what the chosen variant's register budget holds
    if (best_units.empty()) TZ_FAIL(TZ_ERR_UNSUPPORTED, "no Gram plan for Tz=%d", Tz);
    p->gu = best_u;
    TZ_HIP(p->gunits.upload(best_units)); TZ_HIP(p->gunit_ptr.upload(best_ptr));
    p->mfma_gram = (int64_t)best_span; p->mfma_chol = (int64_t)Tz * Tz * Tz / 24; p->mfma_issued = p->mfma_gram * TZ_NWAVES + p->mfma_chol;
  } else {
    p->ksplit = (p->Tz <= TZ_KS_TZ);
    p->chol1 = (p->Tz <= 16);
    if (d->plan_flags & TZ_PLAN_GENERAL_CHOLESKY) p->chol1 = false;
    if (d->plan_flags & TZ_PLAN_ITEM_GRAM) p->ksplit = false;
    if (const char* e = tz_dev_getenv("TZ_CHOL1")) { if (e[0] == '0') p->chol1 = false; }
    if (const char* e = tz_dev_getenv("TZ_KSPLIT")) { if (e[0] == '0') p->ksplit = false; }
    p->hsize = (size_t)p->nquads * TZ_QSTR;
    // nz <= 64: the 128-register variant, h and G x of the rows parked in LDS
    p->lds_bytes = tz_ipm_lds_doubles(p->hsize, 0, Tz, nzp, mip, p->nklist, p->ntheta, p->ksplit ? 1 : 0, p->ntube, p->nell, 1) * sizeof(double);
    if (p->lds_bytes > LDS_MAX) return fail_lds(p->lds_bytes);
    wgs_per_cu = (int)(LDS_MAX / std::max<size_t>(p->lds_bytes, 1));
    bool ttk = false;
    p->ipm_fn = ipm_kernel_for(p->maxr, p->ncg, wgs_per_cu, ttk);
  }
  if (!p->ipm_fn) TZ_FAIL(TZ_ERR_UNSUPPORTED, "this development build (TZ_ONLY_SMALL) carries only the mi <= 256, nz <= 64 kernel");
  {                                                                     // the G x / G'v tables were built for the class decided above
    bool ttk = false;
    (void)ipm_kernel_for(p->maxr, p->ncg, wgs_per_cu, ttk);
    if (ttk != p->tt) TZ_FAIL(TZ_ERR_UNSUPPORTED, "kernel class and table format disagree (development build?)");
  }
  TZ_HIP(hipFuncSetAttribute((const void*)p->ipm_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
  if (const char* e = tz_dev_getenv("TZ_PROF")) { p->prof = (e[0] == '1'); }
  if (p->prof && !TZ_PROFILE) TZ_FAIL(TZ_ERR_INVALID, "TZ_PROF=1 needs the diagnostic build of the library (libtzddpc_hip_prof.so)");
  if (const char* e = tz_dev_getenv("TZ_WARM")) { p->warm_enabled = (e[0] != '0'); }
  if (d->plan_flags & TZ_PLAN_UNFUSED) p->fuse_enabled = false;
  if (const char* e = tz_dev_getenv("TZ_FUSE")) { p->fuse_enabled = (e[0] != '0'); }
  // the tube pass of the fused step keeps |C_K^l e0| (pmax n doubles) in the factor storage, which is free at that point: a short-horizon,
  // large-n problem whose factor is smaller than that runs the four-kernel step instead (tz_tube_kernel has its own scratch)
  if ((size_t)std::max(p->pmax, 1) * p->n > p->hsize) p->fuse_enabled = false;
  // experiment switches (tools/): the same ranges as the setters (tz_problem_set_stopping / _warm_quiet / _warm_push) -- a value
  // that does not parse or lies outside fails the create call instead of silently changing the solver's accuracy
  {
    auto envd = [](const char* name, double lo, bool lo_open, double hi, double& dst) -> bool {
      const char* e = tz_dev_getenv(name);
      if (!e) return true;
      char* end = nullptr;
      const double v = strtod(e, &end);
      if (end == e || *end != '\0' || !(lo_open ? v > lo : v >= lo) || !(v <= hi)) return false;
      dst = v;
      return true;
    };
    double sq = (double)p->shift_quiet;
    if (!envd("TZ_SHIFT_QUIET", 0.0, false, 1e9, sq)) TZ_FAIL(TZ_ERR_INVALID, "TZ_SHIFT_QUIET must be an integer >= 0");
    p->shift_quiet = (int)sq;
    if (!envd("TZ_WARM_FLOOR", 0.0, true, 1e300, p->warm_floor)) TZ_FAIL(TZ_ERR_INVALID, "TZ_WARM_FLOOR must be > 0");
    if (!envd("TZ_WARM_GAIN", 0.0, false, 1e300, p->warm_gain)) TZ_FAIL(TZ_ERR_INVALID, "TZ_WARM_GAIN must be >= 0");
    if (!envd("TZ_WARM_CAP", 0.0, true, 1e300, p->warm_cap) || p->warm_cap < p->warm_floor) TZ_FAIL(TZ_ERR_INVALID, "TZ_WARM_CAP must be >= the floor");
    if (!envd("TZ_MU_FACTOR", 0.0, true, 1.0, p->mu_factor)) TZ_FAIL(TZ_ERR_INVALID, "TZ_MU_FACTOR must be in (0, 1]");
    if (!envd("TZ_RES_FACTOR", 1.0, false, 1e300, p->res_factor)) TZ_FAIL(TZ_ERR_INVALID, "TZ_RES_FACTOR must be >= 1");
    if (!envd("TZ_AFF_THR", 0.0, true, 1.0, p->aff_thr)) TZ_FAIL(TZ_ERR_INVALID, "TZ_AFF_THR must be in (0, 1]");
    if (!envd("TZ_AFF_MU", 0.0, true, 1.0, p->aff_mu)) TZ_FAIL(TZ_ERR_INVALID, "TZ_AFF_MU must be in (0, 1]");
    double sf = p->step_frac;
    if (!envd("TZ_STEP_FRAC", 0.0, true, 1.0, sf) || !(sf < 1.0)) TZ_FAIL(TZ_ERR_INVALID, "TZ_STEP_FRAC must be in (0, 1)");
    p->step_frac = sf;
  }
  if (p->prof) { TZ_HIP(p->prof_buf.alloc(PH_COUNT + 16)); TZ_HIP(hipMemset(p->prof_buf.p, 0, (PH_COUNT + 16) * sizeof(unsigned long long))); }
  TZ_HIP(p->work_buf.alloc(3));
  TZ_HIP(hipMemset(p->work_buf.p, 0, 3 * sizeof(unsigned long long)));
  TZ_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  p->own_stream = true;
  *out = guard.release();
  return TZ_OK;
}

int tz_problem_destroy(tz_problem* p) {
  if (!p) return TZ_OK;
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  for (int k = 0; k < K_COUNT; ++k)
    for (auto& e : p->ev_pool[k]) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return TZ_OK;
}

int tz_problem_set_stream(tz_problem* p, void* stream) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  TZ_HIP(hipSetDevice(p->device));
  if (p->stream) TZ_HIP(hipStreamSynchronize(p->stream));
  if (p->own_stream && p->stream) { TZ_HIP(hipStreamDestroy(p->stream)); p->own_stream = false; }
  if (stream == nullptr) { TZ_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)); p->own_stream = true; }
  else p->stream = (hipStream_t)stream;
  return TZ_OK;
}

int tz_problem_sync(tz_problem* p) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  TZ_HIP(hipSetDevice(p->device));
  TZ_HIP(hipStreamSynchronize(p->stream));
  return TZ_OK;
}

int tz_solve_batch(tz_problem* p, int32_t B, const double* xbar0, const double* e0, double* v, double* xbar,
                   double* cost, int32_t* status, int32_t* iters, uint8_t* active, int mem) {
  if (!p || !xbar0 || !e0 || !v || !xbar || !cost || !status) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (B < 0) TZ_FAIL(TZ_ERR_INVALID, "negative batch size");
  if (B == 0) return TZ_OK;
  TZ_HIP(hipSetDevice(p->device));
  int rc = ensure_workspace(p, B);
  if (rc) return rc;
  p->have_prev = false;
  const size_t bn = (size_t)B * p->n * sizeof(double);
  if (mem == TZ_MEM_DEVICE) {
    return launch_solve(p, B, xbar0, e0, v, xbar, cost, status, iters, active);
  }
  if (mem != TZ_MEM_HOST) TZ_FAIL(TZ_ERR_INVALID, "mem must be TZ_MEM_HOST or TZ_MEM_DEVICE");
  hipStream_t st = p->stream;
  TZ_HIP(hipMemcpyAsync(p->in_x0.p, xbar0, bn, hipMemcpyHostToDevice, st));
  TZ_HIP(hipMemcpyAsync(p->in_e0.p, e0, bn, hipMemcpyHostToDevice, st));
  rc = launch_solve(p, B, p->in_x0.p, p->in_e0.p, p->v.p, p->xbar.p, p->cost.p, p->status.p, p->iters.p, active ? p->active.p : nullptr);
  if (rc) return rc;
  TZ_HIP(hipMemcpyAsync(v, p->v.p, (size_t)B * p->N * p->m * sizeof(double), hipMemcpyDeviceToHost, st));
  TZ_HIP(hipMemcpyAsync(xbar, p->xbar.p, (size_t)B * (p->N + 1) * p->n * sizeof(double), hipMemcpyDeviceToHost, st));
  TZ_HIP(hipMemcpyAsync(cost, p->cost.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
  TZ_HIP(hipMemcpyAsync(status, p->status.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
  if (iters) TZ_HIP(hipMemcpyAsync(iters, p->iters.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
  if (active) TZ_HIP(hipMemcpyAsync(active, p->active.p, (size_t)B * p->nc_rows, hipMemcpyDeviceToHost, st));
  TZ_HIP(hipStreamSynchronize(st));
  return TZ_OK;
}

// Stored start: the first closed-loop step of a trajectory that has no previous solution starts from the solution of ONE reference
// solve (tz_problem_store_start: the caller's point, typically the centre of X0 with e0 = 0) instead of from the cold point.
__global__ void tz_seed_kernel(int B, int nz, int mi, const double* rx, const double* rl, double* x, double* lam, int* prev_status, int* iters, int* shift_state) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)B * nz) x[i] = rx[i % nz];
  if (i < (size_t)B * mi) lam[i] = rl[i % mi];
  if (i < (size_t)B) { prev_status[i] = 0; iters[i] = 0; shift_state[i] = -2; }      // -2: stored start (tz_ipm_kernel: taken unshifted, then the shifted regime)
}
static int seed_warm(tz_problem* p, int B) {
  const size_t total = (size_t)B * std::max(p->nz, p->mi);
  hipLaunchKernelGGL(tz_seed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, p->stream, B, p->nz, p->mi, p->ref_x.p, p->ref_lam.p,
                     p->x.p, p->lam.p, p->prev_status.p, p->iters.p, p->shift_state.p);
  TZ_HIP(hipGetLastError());
  p->have_prev = true; p->prevB = B;
  return TZ_OK;
}
// warm-start state of a closed-loop launch: the previous step's, else the stored start's, else none (cold)
static int closed_loop_warm(tz_problem* p, int B, bool* warm) {
  *warm = p->warm_enabled && p->have_prev && p->prevB == B;
  if (!*warm && p->warm_enabled && p->have_ref) { int rc = seed_warm(p, B); if (rc) return rc; *warm = true; }
  return TZ_OK;
}

int tz_problem_store_start(tz_problem* p, const double* xbar0, const double* e0) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  if (!xbar0 || !e0) { p->have_ref = false; return TZ_OK; }                       // NULL: forget the stored start
  TZ_HIP(hipSetDevice(p->device));
  int rc = ensure_workspace(p, 1);
  if (rc) return rc;
  TZ_HIP(hipMemcpyAsync(p->in_x0.p, xbar0, (size_t)p->n * sizeof(double), hipMemcpyHostToDevice, p->stream));
  TZ_HIP(hipMemcpyAsync(p->in_e0.p, e0, (size_t)p->n * sizeof(double), hipMemcpyHostToDevice, p->stream));
  rc = launch_solve(p, 1, p->in_x0.p, p->in_e0.p, p->v.p, p->xbar.p, p->cost.p, p->status.p, p->iters.p, nullptr);
  if (rc) return rc;
  int st = -1;
  TZ_HIP(hipMemcpyAsync(&st, p->status.p, sizeof(int), hipMemcpyDeviceToHost, p->stream));
  TZ_HIP(hipStreamSynchronize(p->stream));
  if (st != 0) TZ_FAIL(TZ_ERR_INVALID, "the reference point is not solvable (status %d): no start stored", st);
  TZ_HIP(p->ref_x.alloc((size_t)p->nz)); TZ_HIP(p->ref_lam.alloc((size_t)p->mi));
  TZ_HIP(hipMemcpyAsync(p->ref_x.p, p->x.p, (size_t)p->nz * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  TZ_HIP(hipMemcpyAsync(p->ref_lam.p, p->lam.p, (size_t)p->mi * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  TZ_HIP(hipStreamSynchronize(p->stream));
  p->have_ref = true; p->have_prev = false;
  return TZ_OK;
}

int tz_mpc_step(tz_problem* p, int32_t B, double* x, double* xbar, double* e, const double* w,
                const double* A_true, const double* B_true, double* u_out, double* cost, int32_t* status) {
  if (!p || !x || !xbar || !e || !w || !A_true || !B_true || !cost || !status) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (B <= 0) TZ_FAIL(TZ_ERR_INVALID, "batch size must be positive");
  TZ_HIP(hipSetDevice(p->device));
  int rc = ensure_workspace(p, B);
  if (rc) return rc;
  bool warm = false;
  if ((rc = closed_loop_warm(p, B, &warm))) return rc;
  if (p->fuse_enabled) {
    rc = launch_step_fused(p, B, x, xbar, e, w, (size_t)p->n, A_true, B_true, u_out, (size_t)p->m, nullptr, 0, cost, 1, status, nullptr, warm);
    if (rc == TZ_OK) { p->have_prev = true; p->prevB = B; }
    return rc;
  }
  rc = launch_solve(p, B, xbar, e, p->v.p, p->xbar.p, cost, status, p->iters.p, nullptr, 1, warm, true);
  if (rc) return rc;
  p->have_prev = true; p->prevB = B;
  return launch_plant(p, B, A_true, B_true, w, (size_t)p->n, p->v.p, p->xbar.p, status, x, xbar, e,
                      u_out, (size_t)p->m, nullptr, 0, nullptr);
}

int tz_mpc_run(tz_problem* p, int32_t B, int32_t K, double* x, double* xbar, double* e, const double* w,
               const double* A_true, const double* B_true, double* u_out, double* cost, int32_t* status) {
  if (!p || !x || !xbar || !e || !w || !A_true || !B_true || !cost || !status) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (B <= 0 || K <= 0) TZ_FAIL(TZ_ERR_INVALID, "B and K must be positive");
  TZ_HIP(hipSetDevice(p->device));
  int rc = ensure_workspace(p, B);
  if (rc) return rc;
  if (p->fuse_enabled) {                    // all K steps of every trajectory in ONE launch: the state never leaves the workgroup
    bool warm = false;
    if ((rc = closed_loop_warm(p, B, &warm))) return rc;
    rc = launch_step_fused(p, B, x, xbar, e, w, (size_t)p->n, A_true, B_true, u_out, (size_t)p->m, nullptr, 0,
                           cost, 1, p->status.p, status, warm, K, StepStrides{(size_t)B * p->n, 0, 0, 0}, true);   // `status` cleared by the kernel
    if (rc) return rc;
    p->have_prev = true; p->prevB = B;
    return TZ_OK;
  }
  TZ_HIP(hipMemsetAsync(status, 0, (size_t)B * sizeof(int), p->stream));
  for (int t = 0; t < K; ++t) {
    bool warm = false;
    if ((rc = closed_loop_warm(p, B, &warm))) return rc;
    rc = launch_solve(p, B, xbar, e, p->v.p, p->xbar.p, cost, p->status.p, p->iters.p, nullptr, 1, warm, true);
    if (rc) return rc;
    p->have_prev = true; p->prevB = B;
    rc = launch_plant(p, B, A_true, B_true, w + (size_t)t * B * p->n, (size_t)p->n, p->v.p, p->xbar.p, p->status.p, x, xbar, e,
                      u_out, (size_t)p->m, nullptr, 0, status);
    if (rc) return rc;
  }
  return TZ_OK;
}

int tz_simulate_batch(tz_problem* p, int32_t B, int32_t T, const double* x0, const double* noise,
                      const double* A_true, const double* B_true, double* x_traj, double* u_traj,
                      double* cost, int32_t* status, int mem) {
  if (!p || !x0 || !noise || !A_true || !B_true || !x_traj || !u_traj || !status) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (B <= 0 || T <= 0) TZ_FAIL(TZ_ERR_INVALID, "B and T must be positive");
  TZ_HIP(hipSetDevice(p->device));
  int rc = ensure_workspace(p, B);
  if (rc) return rc;
  hipStream_t st = p->stream;
  const int n = p->n, m = p->m;
  p->have_prev = false;
  const bool host = (mem == TZ_MEM_HOST);
  if (!host && mem != TZ_MEM_DEVICE) TZ_FAIL(TZ_ERR_INVALID, "mem must be TZ_MEM_HOST or TZ_MEM_DEVICE");
  const double *dA = A_true, *dB = B_true, *dnoise = noise;
  double *dx = x_traj, *du = u_traj, *dcost = cost;
  if (host) {
    TZ_HIP(p->plantA.upload(A_true, (size_t)n * n)); TZ_HIP(p->plantB.upload(B_true, (size_t)n * m));
    TZ_HIP(p->noise.upload(noise, (size_t)B * T * n));
    TZ_HIP(p->xtraj.alloc((size_t)B * (T + 1) * n)); TZ_HIP(p->utraj.alloc((size_t)B * T * m));
    TZ_HIP(p->costtraj.alloc((size_t)B * T));
    dA = p->plantA.p; dB = p->plantB.p; dnoise = p->noise.p; dx = p->xtraj.p; du = p->utraj.p; dcost = p->costtraj.p;
    TZ_HIP(hipMemcpy2DAsync(dx, (size_t)(T + 1) * n * sizeof(double), x0, (size_t)n * sizeof(double), (size_t)n * sizeof(double), B, hipMemcpyHostToDevice, st));
    TZ_HIP(hipMemcpyAsync(p->st_x.p, x0, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice, st));
  } else {
    if (!cost) { TZ_HIP(p->costtraj.alloc((size_t)B * T)); dcost = p->costtraj.p; }
    TZ_HIP(hipMemcpy2DAsync(dx, (size_t)(T + 1) * n * sizeof(double), x0, (size_t)n * sizeof(double), (size_t)n * sizeof(double), B, hipMemcpyDeviceToDevice, st));
    TZ_HIP(hipMemcpyAsync(p->st_x.p, x0, (size_t)B * n * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  TZ_HIP(hipMemcpyAsync(p->st_xbar.p, p->st_x.p, (size_t)B * n * sizeof(double), hipMemcpyDeviceToDevice, st));   // xbar = x0 (:69)
  TZ_HIP(hipMemsetAsync(p->st_e.p, 0, (size_t)B * n * sizeof(double), st));                                        // e = 0   (:70)
  TZ_HIP(hipMemsetAsync(p->sticky.p, 0, (size_t)B * sizeof(int), st));
  bool warm0 = false;                                                     // a fresh loop: the stored start if there is one, else cold
  if ((rc = closed_loop_warm(p, B, &warm0))) return rc;
  if (p->fuse_enabled) {
    rc = launch_step_fused(p, B, p->st_x.p, p->st_xbar.p, p->st_e.p, dnoise, (size_t)T * n, dA, dB,
                           du, (size_t)T * m, dx + n, (size_t)(T + 1) * n,
                           dcost, (size_t)T, p->status.p, p->sticky.p, warm0, T, StepStrides{(size_t)n, (size_t)m, (size_t)n, 1});
    if (rc) return rc;
  }
  for (int t = 0; t < T && !p->fuse_enabled; ++t) {
    rc = launch_solve(p, B, p->st_xbar.p, p->st_e.p, p->v.p, p->xbar.p, dcost + t, p->status.p, p->iters.p, nullptr, (size_t)T, p->warm_enabled && (t > 0 || warm0), true);
    if (rc) return rc;
    rc = launch_plant(p, B, dA, dB, dnoise + (size_t)t * n, (size_t)T * n, p->v.p, p->xbar.p, p->status.p,
                      p->st_x.p, p->st_xbar.p, p->st_e.p, du + (size_t)t * m, (size_t)T * m,
                      dx + (size_t)(t + 1) * n, (size_t)(T + 1) * n, p->sticky.p);
    if (rc) return rc;
  }
  if (host) {
    TZ_HIP(hipMemcpyAsync(x_traj, dx, (size_t)B * (T + 1) * n * sizeof(double), hipMemcpyDeviceToHost, st));
    TZ_HIP(hipMemcpyAsync(u_traj, du, (size_t)B * T * m * sizeof(double), hipMemcpyDeviceToHost, st));
    if (cost) TZ_HIP(hipMemcpyAsync(cost, dcost, (size_t)B * T * sizeof(double), hipMemcpyDeviceToHost, st));
    TZ_HIP(hipMemcpyAsync(status, p->sticky.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    TZ_HIP(hipStreamSynchronize(st));
  } else {
    TZ_HIP(hipMemcpyAsync(status, p->sticky.p, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, st));
  }
  return TZ_OK;
}

int tz_problem_set_warm_shift(tz_problem* p, int32_t policy) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  if (policy < 0) TZ_FAIL(TZ_ERR_INVALID, "policy must be >= 0");
  if (policy != 0 && !p->have_shift) TZ_FAIL(TZ_ERR_INVALID, "the problem was created without shift maps");
  p->shift_policy = policy;
  return TZ_OK;
}

int tz_problem_set_stopping(tz_problem* p, double res_factor, double mu_factor) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  if (!(res_factor >= 1.0) || !(mu_factor > 0.0) || !(mu_factor <= 1.0)) TZ_FAIL(TZ_ERR_INVALID, "need res_factor >= 1 and 0 < mu_factor <= 1");
  p->res_factor = res_factor; p->mu_factor = mu_factor;
  return TZ_OK;
}

int tz_problem_set_warm_quiet(tz_problem* p, int32_t quiet_steps) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  if (quiet_steps < 0) TZ_FAIL(TZ_ERR_INVALID, "quiet_steps must be >= 0");
  p->shift_quiet = quiet_steps;
  return TZ_OK;
}

int tz_problem_set_warm_push(tz_problem* p, double floor, double gain, double cap) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  if (!(floor > 0.0) || !(gain >= 0.0) || !(cap >= floor)) TZ_FAIL(TZ_ERR_INVALID, "floor must be positive, gain non-negative, cap >= floor");
  p->warm_floor = floor; p->warm_gain = gain; p->warm_cap = cap;
  return TZ_OK;
}

int tz_problem_reset_warm(tz_problem* p) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  p->have_prev = false;
  if (p->Bcap > 0) {
    TZ_HIP(hipSetDevice(p->device));
    TZ_HIP(hipMemsetAsync(p->shift_state.p, 0, (size_t)p->Bcap * sizeof(int), p->stream));
  }
  return TZ_OK;
}

int tz_timing_enable(tz_problem* p, int enable) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  TZ_HIP(hipSetDevice(p->device));
  if (p->timing) drain_timing(p);
  p->timing = enable != 0;
  if (p->timing)                                   // events are created here, not inside a timed region
    for (int k = 0; k < K_COUNT; ++k)
      while (p->ev_pool[k].size() < 512) {
        hipEvent_t a, b;
        TZ_HIP(hipEventCreate(&a)); TZ_HIP(hipEventCreate(&b));
        p->ev_pool[k].push_back({a, b});
      }
  TZ_HIP(hipStreamSynchronize(p->stream));
  TZ_HIP(hipMemset(p->work_buf.p, 0, 3 * sizeof(unsigned long long)));
  for (int k = 0; k < K_COUNT; ++k) { p->t_ms[k] = 0; p->t_count[k] = 0; }
  return TZ_OK;
}

int tz_timing_get(tz_problem* p, int kernel, double* total_ms, int64_t* launches) {
  if (!p || kernel < 0 || kernel >= K_COUNT || !total_ms || !launches) TZ_FAIL(TZ_ERR_INVALID, "bad argument");
  TZ_HIP(hipSetDevice(p->device));
  drain_timing(p);
  *total_ms = p->t_ms[kernel]; *launches = p->t_count[kernel];
  return TZ_OK;
}

int tz_ipm_work_get(tz_problem* p, int64_t* factorizations, int64_t* trajectory_solves, int64_t* max_factorizations_one_trajectory) {
  if (!p || !factorizations || !trajectory_solves) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  TZ_HIP(hipSetDevice(p->device));
  TZ_HIP(hipStreamSynchronize(p->stream));
  unsigned long long h[3];
  TZ_HIP(hipMemcpy(h, p->work_buf.p, sizeof(h), hipMemcpyDeviceToHost));
  *factorizations = (int64_t)h[0]; *trajectory_solves = (int64_t)h[1];
  if (max_factorizations_one_trajectory) *max_factorizations_one_trajectory = (int64_t)h[2];
  return TZ_OK;
}

int tz_ipm_plan_info(tz_problem* p, int64_t* mfma_gram_per_iter, int64_t* mfma_chol_per_iter, int64_t* mfma_issued_per_iter,
                     int64_t* lds_bytes, int64_t* patch_bytes) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  if (mfma_gram_per_iter) *mfma_gram_per_iter = p->mfma_gram;
  if (mfma_chol_per_iter) *mfma_chol_per_iter = p->mfma_chol;
  if (mfma_issued_per_iter) *mfma_issued_per_iter = p->mfma_issued;
  if (lds_bytes) *lds_bytes = (int64_t)p->lds_bytes;
  if (patch_bytes) *patch_bytes = (int64_t)p->Gp.n * 8;
  return TZ_OK;
}

int tz_problem_plan_get(tz_problem* p, int32_t* fused, int32_t* one_wave_cholesky, int32_t* superstep_gram, int32_t* staircase, int32_t* toeplitz) {
  if (!p) TZ_FAIL(TZ_ERR_INVALID, "null problem");
  if (fused) *fused = p->fuse_enabled ? 1 : 0;
  if (one_wave_cholesky) *one_wave_cholesky = p->chol1 ? 1 : 0;
  if (superstep_gram) *superstep_gram = p->ksplit ? 1 : 0;
  if (staircase) *staircase = p->staircase ? 1 : 0;
  if (toeplitz) *toeplitz = p->toeplitz ? 1 : 0;
  return TZ_OK;
}

int tz_debug_fetch(tz_problem* p, int32_t b, int what, double* out, int32_t capacity) {
  if (!p || !out) TZ_FAIL(TZ_ERR_INVALID, "null argument");
  if (b < 0 || b >= p->lastB) TZ_FAIL(TZ_ERR_INVALID, "trajectory %d outside the last batch (%d)", b, p->lastB);
  TZ_HIP(hipSetDevice(p->device));
  TZ_HIP(hipStreamSynchronize(p->stream));
  const double* src = nullptr; int len = 0;
  switch (what) {
    case 0: src = p->theta.p + (size_t)b * p->ntheta; len = p->ntheta; break;
    case 1: src = p->qv.p + (size_t)b * p->nz; len = p->nz; break;
    case 2: src = p->hv.p + (size_t)b * p->mi; len = p->mi; break;
    case 3: src = p->x.p + (size_t)b * p->nz; len = p->nz; break;
    case 4: src = p->s.p + (size_t)b * p->mi; len = p->mi; break;
    case 5: src = p->lam.p + (size_t)b * p->mi; len = p->mi; break;
    case 6: {   // diagnostic build: per-phase cycle sums of workgroup 0 (as doubles)
      if (!p->prof) TZ_FAIL(TZ_ERR_INVALID, "profiling build not enabled (TZ_PROF=1 at problem creation)");
      unsigned long long h[PH_COUNT + 16];
      TZ_HIP(hipMemcpy(h, p->prof_buf.p, sizeof(h), hipMemcpyDeviceToHost));
      if (capacity < PH_COUNT + 16) TZ_FAIL(TZ_ERR_INVALID, "capacity too small");
      for (int i = 0; i < PH_COUNT + 16; ++i) out[i] = (double)h[i];
      TZ_HIP(hipMemset(p->prof_buf.p + 32, 0, 6 * sizeof(unsigned long long)));      // slots 32..37: sums of other waves (atomics), restart
      return PH_COUNT + 16;
    }
    case 7: {   // interior-point iterations of every trajectory of the last launch (b ignored)
      if (capacity < p->lastB) TZ_FAIL(TZ_ERR_INVALID, "capacity %d < %d", capacity, p->lastB);
      std::vector<int> h((size_t)p->lastB);
      TZ_HIP(hipMemcpy(h.data(), p->iters.p, h.size() * sizeof(int), hipMemcpyDeviceToHost));
      for (int i = 0; i < p->lastB; ++i) out[i] = (double)h[i];
      return p->lastB;
    }
    default: TZ_FAIL(TZ_ERR_INVALID, "unknown debug item %d", what);
  }
  if (capacity < len) TZ_FAIL(TZ_ERR_INVALID, "capacity %d < %d", capacity, len);
  TZ_HIP(hipMemcpy(out, src, (size_t)len * sizeof(double), hipMemcpyDeviceToHost));
  const std::vector<int>* perm = (what == 1 || what == 3) ? &p->permc : ((what == 2 || what == 4 || what == 5) ? &p->permr : nullptr);
  if (perm && !perm->empty()) {                       // device order -> the caller's order
    std::vector<double> tmp(out, out + len);
    for (int i = 0; i < len; ++i) out[(*perm)[i]] = tmp[i];
  }
  return len;
}

}  // extern "C"
