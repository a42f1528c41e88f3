/*
 * tz_oracle.c -- plain-C restatement of the per-step numeric path.  ORACLE / TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * (tzddpc_amd/) never does.  PARITY UNPINNED: the reference has no fixtures for this path and cannot
 * be run here (oracle/__init__.py).
 *
 * What it restates (reference rssalessio/TZDDPC):
 *   tube propagation from e0            tzddpc/tzddpc.py:172-181, 191-192 (numeric content)
 *   parameter application               tzddpc/tzddpc.py:364-365
 *   the conic solve                     tzddpc/tzddpc.py:367  (an interior-point solver in the reference too)
 *   returned v, xbar, objective         tzddpc/tzddpc.py:377
 *   plant / error update                examples/1.double_integrator_sim.py:83-87
 *
 * Input is the UNSCALED two-sided parametric QP  min 1/2 z'Pz + q(theta)'z,  l(theta) <= A z <= u(theta);
 * this file does its own equilibration, one-sided conversion, dense Cholesky (no blocking, no MFMA
 * structure) and Mehrotra predictor-corrector, so agreement with the HIP kernels is a real check.
 * OpenMP over trajectories (threads argument); scalar code otherwise.
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct tzo_desc {
  int32_t n, m, N, nz, nc, ntheta, npar;
  const double *P, *A;            /* nz x nz, nc x nz */
  const double *q0, *Qt;          /* nz, nz x ntheta */
  const double *l0, *Lt, *u0, *Ut;/* nc, nc x ntheta (entries of l0/u0 may be -/+inf) */
  const double *f0, *Ft, *pl, *pu;/* parameter-only rows */
  double r0; const double *r1, *R2;
  const double *Phi, *Gam;        /* (N+1)n x n, (N+1)n x N m */
  const double *CK, *DK, *K;      /* n x n, n x n, m x n */
  int32_t pmax; const double *absCK, *absKCK; const int32_t *power;
  int32_t max_iter; double tol, reg, step_frac;
  double warm_floor, warm_gain, warm_cap;   /* closed-loop warm start (tzo_simulate_batch): see ipm() */
  double mu_tol;                  /* complementarity target (<= tol) */
  double res_tol;                 /* residual tolerance of the stopping test (>= tol) */
  double aff_thr, aff_mu;         /* predictor step taken as the step when it is (nearly) full and leaves mu_aff <= aff_mu mu */
  const int32_t *shift_var, *shift_row;   /* receding-horizon shift of the warm start: source variable (nz) / two-sided row (nc) */
  int32_t shift_policy;           /* 0 never, 1 always, k >= 2: after a step of >= k iterations and while the shifted steps that
                                   * follow take one iteration (tz_problem_set_warm_shift) */
  int32_t shift_quiet;            /* k >= 2: the shifted regime is left after this many one-iteration shifted steps in a row (0: never) */
  const double* start_xbar0;      /* n, or NULL: stored start (tz_problem_store_start of the device library) -- closed loops begin from the
                                   * solution of ONE cold solve at (start_xbar0, e0 = 0) instead of from the cold point */
} tzo_desc;

typedef struct {
  int nz, mi;
  double *P, *G, *D, *E; double c;
  int *row, *sgn;                 /* one-sided row -> (two-sided row, +1 upper / -1 lower) */
  int *srow;                      /* one-sided source row of the shifted warm start (NULL without shift maps) */
} setup_t;

static void ruiz(int nz, int mi, double* P, double* G, double* D, double* E, int iters) {
  for (int i = 0; i < nz; ++i) D[i] = 1.0;
  for (int r = 0; r < mi; ++r) E[r] = 1.0;
  double* d = (double*)malloc(sizeof(double) * nz);
  double* e = (double*)malloc(sizeof(double) * mi);
  for (int it = 0; it < iters; ++it) {
    for (int c = 0; c < nz; ++c) {
      double mx = 0;
      for (int r = 0; r < nz; ++r) mx = fmax(mx, fabs(P[r * nz + c]));
      for (int r = 0; r < mi; ++r) mx = fmax(mx, fabs(G[r * nz + c]));
      d[c] = 1.0 / sqrt(mx < 1e-8 ? 1.0 : mx);
    }
    for (int r = 0; r < mi; ++r) {
      double mx = 0;
      for (int c = 0; c < nz; ++c) mx = fmax(mx, fabs(G[r * nz + c]));
      e[r] = 1.0 / sqrt(mx < 1e-8 ? 1.0 : mx);
    }
    for (int r = 0; r < nz; ++r) for (int c = 0; c < nz; ++c) P[r * nz + c] *= d[r] * d[c];
    for (int r = 0; r < mi; ++r) for (int c = 0; c < nz; ++c) G[r * nz + c] *= e[r] * d[c];
    for (int c = 0; c < nz; ++c) D[c] *= d[c];
    for (int r = 0; r < mi; ++r) E[r] *= e[r];
  }
  free(d); free(e);
}

static setup_t* make_setup(const tzo_desc* d) {
  setup_t* S = (setup_t*)calloc(1, sizeof(setup_t));
  int nz = d->nz, nc = d->nc, mi = 0;
  for (int r = 0; r < nc; ++r) { if (isfinite(d->u0[r])) mi++; if (isfinite(d->l0[r])) mi++; }
  S->nz = nz; S->mi = mi;
  S->P = (double*)malloc(sizeof(double) * nz * nz); memcpy(S->P, d->P, sizeof(double) * nz * nz);
  S->G = (double*)malloc(sizeof(double) * mi * nz);
  S->D = (double*)malloc(sizeof(double) * nz); S->E = (double*)malloc(sizeof(double) * mi);
  S->row = (int*)malloc(sizeof(int) * mi); S->sgn = (int*)malloc(sizeof(int) * mi);
  int k = 0;
  for (int r = 0; r < nc; ++r) if (isfinite(d->u0[r])) { S->row[k] = r; S->sgn[k] = 1; for (int c = 0; c < nz; ++c) S->G[k * nz + c] = d->A[r * nz + c]; k++; }
  for (int r = 0; r < nc; ++r) if (isfinite(d->l0[r])) { S->row[k] = r; S->sgn[k] = -1; for (int c = 0; c < nz; ++c) S->G[k * nz + c] = -d->A[r * nz + c]; k++; }
  ruiz(nz, mi, S->P, S->G, S->D, S->E, 15);
  double pn = 0, qn = 0;
  for (int i = 0; i < nz * nz; ++i) pn = fmax(pn, fabs(S->P[i]));
  for (int c = 0; c < nz; ++c) {
    double a = fabs(d->q0[c]);
    for (int t = 0; t < d->ntheta; ++t) a += fabs(d->Qt[c * d->ntheta + t]);
    qn = fmax(qn, S->D[c] * a);
  }
  S->c = 1.0 / fmax(fmax(pn, qn), 1e-300);
  for (int i = 0; i < nz * nz; ++i) S->P[i] *= S->c;
  S->srow = NULL;
  if (d->shift_row && d->shift_var) {
    S->srow = (int*)malloc(sizeof(int) * mi);
    for (int a = 0; a < mi; ++a) {
      int target = d->shift_row[S->row[a]]; S->srow[a] = a;
      for (int b2 = 0; b2 < mi; ++b2) if (S->row[b2] == target && S->sgn[b2] == S->sgn[a]) { S->srow[a] = b2; break; }
    }
  }
  return S;
}

static void free_setup(setup_t* S) { free(S->srow); free(S->P); free(S->G); free(S->D); free(S->E); free(S->row); free(S->sgn); free(S); }

/* theta = [xbar0 | |xbar0| | (c_k, rho^x_k, rho^u_k)_k]: literal restatement of the collapsed recursion */
static void tube_theta(const tzo_desc* d, const double* xbar0, const double* e0, double* th, double* ws) {
  int n = d->n, m = d->m, pm = d->pmax;
  double* c = ws;                       /* (pm+1) x n */
  double* beta = c + (pm + 1) * n;      /* pm x n */
  double* rx = beta + (pm > 0 ? pm : 1) * n;   /* (pm+1) x n */
  double* ru = rx + (pm + 1) * n;       /* (pm+1) x m */
  for (int i = 0; i < n; ++i) { th[i] = xbar0[i]; th[n + i] = fabs(xbar0[i]); c[i] = e0[i]; rx[i] = 0; }
  for (int j = 0; j < m; ++j) ru[j] = 0;
  for (int p = 0; p < pm; ++p) {
    for (int i = 0; i < n; ++i) {
      double a = 0; for (int j = 0; j < n; ++j) a += d->DK[i * n + j] * (fabs(c[p * n + j]) + rx[p * n + j]);
      beta[p * n + i] = a;
      double b = 0; for (int j = 0; j < n; ++j) b += d->CK[i * n + j] * c[p * n + j];
      c[(p + 1) * n + i] = b;
    }
    for (int i = 0; i < n; ++i) {
      double a = 0;
      for (int l = 0; l <= p; ++l) for (int j = 0; j < n; ++j) a += d->absCK[((p - l) * n + i) * n + j] * beta[l * n + j];
      rx[(p + 1) * n + i] = a;
    }
    for (int i = 0; i < m; ++i) {
      double a = 0;
      for (int l = 0; l <= p; ++l) for (int j = 0; j < n; ++j) a += d->absKCK[((p - l) * m + i) * n + j] * beta[l * n + j];
      ru[(p + 1) * m + i] = a;
    }
  }
  int blk = 2 * n + m;
  for (int k = 0; k < d->N; ++k) {
    int p = d->power[k];
    double* dst = th + 2 * n + k * blk;
    for (int i = 0; i < n; ++i) { dst[i] = c[p * n + i]; dst[n + i] = rx[p * n + i]; }
    for (int j = 0; j < m; ++j) dst[2 * n + j] = ru[p * m + j];
  }
}

static int cholesky(int n, double* H) {   /* lower, in place, unblocked */
  for (int j = 0; j < n; ++j) {
    double dg = H[j * n + j];
    for (int k = 0; k < j; ++k) dg -= H[j * n + k] * H[j * n + k];
    if (!(dg > 0)) return 0;
    dg = sqrt(dg); H[j * n + j] = dg;
    for (int i = j + 1; i < n; ++i) {
      double a = H[i * n + j];
      for (int k = 0; k < j; ++k) a -= H[i * n + k] * H[j * n + k];
      H[i * n + j] = a / dg;
    }
  }
  return 1;
}
static void chol_solve(int n, const double* L, double* x) {
  for (int i = 0; i < n; ++i) { double a = x[i]; for (int k = 0; k < i; ++k) a -= L[i * n + k] * x[k]; x[i] = a / L[i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { double a = x[i]; for (int k = i + 1; k < n; ++k) a -= L[k * n + i] * x[k]; x[i] = a / L[i * n + i]; }
}

static void form_H(const setup_t* S, const double* w, double reg, double* H, double* GW) {
  int nz = S->nz, mi = S->mi;
  for (int r = 0; r < mi; ++r) for (int c = 0; c < nz; ++c) GW[r * nz + c] = S->G[r * nz + c] * w[r];
  for (int i = 0; i < nz; ++i)
    for (int j = 0; j <= i; ++j) {
      double a = S->P[i * nz + j];
      for (int r = 0; r < mi; ++r) a += GW[r * nz + i] * S->G[r * nz + j];
      H[i * nz + j] = a;
    }
  for (int i = 0; i < nz; ++i) H[i * nz + i] += reg;
}

static double max_step(int n, const double* v, const double* dv) {
  double a = 1e300;
  for (int i = 0; i < n; ++i) if (dv[i] < 0) a = fmin(a, -v[i] / dv[i]);
  return a;
}

/* TZO_TRACE=1: per-iteration mu / residuals / step lengths on stderr (how the warm start of round 2 was found); read once */
static int tzo_trace(void) { static int t = -1; if (t < 0) t = getenv("TZO_TRACE") != NULL; return t; }

/* what a solved step leaves for the next one (see `carry` in ipm): a breakdown -> its final level and gate; no shift at all -> forget */
#define TZO_RLEV_GATE 1e6   /* as TZ_RLEV_GATE of the device kernel */
#define IPM_DONE() do { if (carry && regx0 == 0.0) { if (brk > 0.0) { carry[0] = regx; carry[1] = brk; } else if (regx == 0.0) { carry[0] = 0.0; carry[1] = 0.0; } } } while (0)
/* status: 0 solved, 1 max_iter, 2 numerical, 3 infeasible */
/* warm != 0: x / lam hold the previous closed-loop step's solution of this trajectory; the slacks are re-derived for the
 * new h and (s, lam) pushed into the cone: sig = min(max(warm_floor, warm_gain * largest violation of the new rows), warm_cap), s >= sig,
 * lam >= sig^2 / s. */
#define TZO_SEED_VIOL_MAX 0.1
static int ipm(const tzo_desc* d, const setup_t* S, const double* q, const double* h, double* x, double* s, double* lam, int* iters, double* wk, int warm, double regx0, double* carry) {
  /* warm == 2: as warm == 1 and gx (G x of the starting point) is still valid in the work area from the previous step;
   * warm == 3: the previous (x, lambda) moved one step along the horizon first (values move unscaled, hence the D / E ratios) */
  int nz = S->nz, mi = S->mi;
  double* H = wk; double* GW = H + nz * nz;
  double* w = GW + mi * nz; double* rd = w + mi; double* rp = rd + nz; double* r1 = rp + mi;
  double* dx = r1 + nz; double* ds = dx + nz; double* dl = ds + mi; double* t1 = dl + mi; double* gx = t1 + mi; double* gdx = gx + mi; double* rc = gdx + mi;
  if (warm) {
    if (warm == 3) {
      for (int c = 0; c < nz; ++c) { int sc = d->shift_var[c]; dx[c] = x[sc] * (S->D[sc] / S->D[c]); }
      for (int c = 0; c < nz; ++c) x[c] = dx[c];
      for (int r = 0; r < mi; ++r) { int sr = S->srow[r]; dl[r] = lam[sr] * (S->E[sr] / S->E[r]); }
      for (int r = 0; r < mi; ++r) lam[r] = dl[r];
    }
    double viol = 0;
    for (int r = 0; r < mi; ++r) {
      if (warm != 2) { double a = 0; for (int c = 0; c < nz; ++c) a += S->G[r * nz + c] * x[c]; gx[r] = a; }
      viol = fmax(viol, gx[r] - h[r]);
    }
    /* warm == 4: the stored start (one reference solve shared by all trajectories): as warm == 1, but a trajectory whose new rows it
     * violates by more than TZO_SEED_VIOL_MAX (equilibrated units) is far from the reference point and starts cold -- measured on
     * jittered starts of the double integrator: below 0.1 the stored start needs 3-9 iterations, above 0.4 it needs 13-21, more than
     * the 13 of a cold start */
    if (warm == 4 && viol > TZO_SEED_VIOL_MAX) warm = 0;
    if (warm) {
    double sig = fmin(fmax(d->warm_floor, d->warm_gain * viol), d->warm_cap);
    if (tzo_trace()) fprintf(stderr, "  warm: viol %.3e sig %.3e mode %d\n", viol, sig, warm);
    /* slack at least sig; multiplier at least sig^2 / s: the pair is pushed onto the central path of mu = sig^2 where it was
     * below it, and an inactive row (large slack, multiplier ~ 0) keeps a multiplier ~ 0 instead of being lifted to sig */
    for (int r = 0; r < mi; ++r) { s[r] = fmax(h[r] - gx[r], sig); lam[r] = fmax(lam[r], sig * sig / s[r]); }
    }
  }
  if (!warm) {
  for (int r = 0; r < mi; ++r) w[r] = 1.0;
  form_H(S, w, d->reg, H, GW);
  if (!cholesky(nz, H)) return 2;
  for (int c = 0; c < nz; ++c) { double a = -q[c]; for (int r = 0; r < mi; ++r) a += S->G[r * nz + c] * h[r]; x[c] = a; }
  chol_solve(nz, H, x);
  double rmin = 1e300;
  for (int r = 0; r < mi; ++r) { double a = 0; for (int c = 0; c < nz; ++c) a += S->G[r * nz + c] * x[c]; gx[r] = a; rmin = fmin(rmin, h[r] - a); }
  double shift = rmin <= 1e-8 ? fmax(0.0, 1.0 - rmin) : 0.0;
  for (int r = 0; r < mi; ++r) { s[r] = h[r] - gx[r] + shift; lam[r] = 1.0; }
  }
  double scd = 0, scp = 0;
  for (int c = 0; c < nz; ++c) scd = fmax(scd, fabs(q[c]));
  for (int r = 0; r < mi; ++r) scp = fmax(scp, fabs(h[r]));
  scd += 1.0; scp += 1.0;
  int it;
  double regx = regx0;               /* diagonal shift of the Newton matrix: 0, raised when a factorisation breaks down; the retry starts at 1e-6 */
  /* carry (closed loops): [0] the shift the previous solved step ended with, [1] TZO_RLEV_GATE x the complementarity at which it first broke down.
   * A warm-started step takes min(carry[0], 1e-6) BEFORE its factorisation fails again, from the iteration on whose mu is below
   * carry[1] (as the device, TZ_RLEV_CARRY_MAX); brk: that multiple of mu at the first breakdown of this solve */
  double brk = 0.0;
  const double carry_lvl = (carry && warm) ? carry[0] : 0.0, carry_gate = (carry && warm) ? carry[1] : 0.0;
  for (it = 0; it < d->max_iter; ++it) {
    double nrd = 0, nrp = 0, mu = 0;
    for (int c = 0; c < nz; ++c) {
      double a = q[c];
      for (int k = 0; k < nz; ++k) a += S->P[c * nz + k] * x[k];
      for (int r = 0; r < mi; ++r) a += S->G[r * nz + c] * lam[r];
      rd[c] = a; nrd = fmax(nrd, fabs(a));
    }
    for (int r = 0; r < mi; ++r) { rp[r] = gx[r] + s[r] - h[r]; nrp = fmax(nrp, fabs(rp[r])); mu += s[r] * lam[r]; }
    mu /= mi; nrd /= scd; nrp /= scp;
    *iters = it;
    if (tzo_trace()) fprintf(stderr, "  it %d mu %.3e rd %.3e rp %.3e warm %d\n", it, mu, nrd, nrp, warm);
    if (nrd <= d->res_tol && nrp <= d->res_tol && mu <= d->mu_tol) { IPM_DONE(); return 0; }
    if (mu <= 1e-3 * d->mu_tol && !(it == 0 && warm))       /* mu collapsed before the residuals: numerical (a warm start may BEGIN */
      { IPM_DONE(); return (nrd <= 1e3 * d->tol && nrp <= 1e3 * d->tol) ? 0 : 2; }   /* there: its first Newton step is what removes the residuals) */
    if (!(mu == mu) || !(nrd == nrd) || mu > 1e200) return 2;
    for (int r = 0; r < mi; ++r) w[r] = lam[r] / s[r];
    if (regx == 0.0 && carry_lvl > 0.0 && mu <= carry_gate) regx = fmin(carry_lvl, 1e-6);
    /* Newton matrix; a factorisation that breaks down (degenerate problems late in the solve: the weights of active and inactive
     * rows are 1e18 apart and H loses definiteness in rounding) raises the diagonal shift, 1e-9, 1e-6, 1e-3, 1 -- kept for the rest
     * of the solve -- and the iteration is repeated from the same point (it counts as an iteration, as on the device).  A shifted H
     * only damps the Newton step (proximal term); the residuals are always exact. */
    form_H(S, w, d->reg + regx, H, GW);
    if (!cholesky(nz, H)) {
      if (regx >= 1.0 || (regx0 == 0.0 && getenv("TZO_NO_ESCALATE"))) return 2;     /* the switch: test of the retry path alone */
      if (brk == 0.0) brk = TZO_RLEV_GATE * mu;
      regx = regx == 0.0 ? 1e-9 : regx * 1e3;
      if (tzo_trace()) fprintf(stderr, "     factorisation failed: diagonal shift %.1e\n", regx);
      continue;
    }
    /* predictor */
    for (int r = 0; r < mi; ++r) t1[r] = w[r] * rp[r] - lam[r];
    for (int c = 0; c < nz; ++c) { double a = -rd[c]; for (int r = 0; r < mi; ++r) a -= S->G[r * nz + c] * t1[r]; dx[c] = a; }
    chol_solve(nz, H, dx);
    for (int r = 0; r < mi; ++r) { double a = 0; for (int c = 0; c < nz; ++c) a += S->G[r * nz + c] * dx[c]; ds[r] = -rp[r] - a; dl[r] = -lam[r] - w[r] * ds[r]; }
    double ap = fmin(1.0, max_step(mi, s, ds)), ad = fmin(1.0, max_step(mi, lam, dl));
    double muaff = 0;
    for (int r = 0; r < mi; ++r) muaff += (s[r] + ap * ds[r]) * (lam[r] + ad * dl[r]);
    muaff /= mi;
    if (fmin(ap, ad) >= d->aff_thr && muaff <= d->aff_mu * mu) {      /* no corrector: the Newton step itself */
      double alphaA = fmin(1.0, d->step_frac * fmin(max_step(mi, s, ds), max_step(mi, lam, dl)));
      for (int c = 0; c < nz; ++c) x[c] += alphaA * dx[c];
      for (int r = 0; r < mi; ++r) { s[r] += alphaA * ds[r]; lam[r] += alphaA * dl[r]; gx[r] += alphaA * (-rp[r] - ds[r]); }
      continue;
    }
    double sigma = muaff / mu; sigma = sigma * sigma * sigma;
    if (tzo_trace()) fprintf(stderr, "     affine ap %.4f ad %.4f muaff/mu %.3e sigma %.3e\n", ap, ad, muaff / mu, sigma);
    /* corrector */
    for (int r = 0; r < mi; ++r) { rc[r] = s[r] * lam[r] + ds[r] * dl[r] - sigma * mu; t1[r] = (-rc[r] + lam[r] * rp[r]) / s[r]; }
    for (int c = 0; c < nz; ++c) { double a = -rd[c]; for (int r = 0; r < mi; ++r) a -= S->G[r * nz + c] * t1[r]; dx[c] = a; }
    chol_solve(nz, H, dx);
    for (int r = 0; r < mi; ++r) { double a = 0; for (int c = 0; c < nz; ++c) a += S->G[r * nz + c] * dx[c]; gdx[r] = a; ds[r] = -rp[r] - a; dl[r] = (-rc[r] - lam[r] * ds[r]) / s[r]; }
    double alpha = fmin(1.0, d->step_frac * fmin(max_step(mi, s, ds), max_step(mi, lam, dl)));
    if (tzo_trace()) fprintf(stderr, "     corrected step: ap %.4f ad %.4f\n", max_step(mi, s, ds), max_step(mi, lam, dl));
    for (int c = 0; c < nz; ++c) x[c] += alpha * dx[c];
    for (int r = 0; r < mi; ++r) { s[r] += alpha * ds[r]; lam[r] += alpha * dl[r]; gx[r] += alpha * gdx[r]; }
  }
  *iters = it;
  return 1;
}

/* Certificate of primal infeasibility carried by the multipliers of a failed solve: y = lam / max(lam) >= 0 with G'y ~ 0 and
 * h'y < 0 (Farkas).  Only then is the failure reported as TZ_INFEASIBLE (the reference raises 'Problem is unbounded' for an
 * infeasible problem, tzddpc/tzddpc.py:374-375); every other failure stays max-iter / numerical. */
static int farkas(const setup_t* S, const double* h, const double* lam) {
  int nz = S->nz, mi = S->mi;
  double lm = 0;
  for (int r = 0; r < mi; ++r) lm = fmax(lm, lam[r]);
  if (!(lm > 0) || !isfinite(lm)) return 0;
  double hy = 0, g = 0;
  for (int r = 0; r < mi; ++r) hy += h[r] * (lam[r] / lm);
  for (int c = 0; c < nz; ++c) { double a = 0; for (int r = 0; r < mi; ++r) a += S->G[r * nz + c] * (lam[r] / lm); g = fmax(g, fabs(a)); }
  return hy < -1e-6 && g <= 1e-6 * fmax(1.0, -hy);
}

static size_t work_doubles(const tzo_desc* d, const setup_t* S) {
  size_t nz = S->nz, mi = S->mi;
  return nz * nz + mi * nz + 9 * mi + 4 * nz /* ipm */ + d->ntheta + nz + 3 * mi /* theta, q, h, s, lam */ + nz
         + (size_t)(d->pmax + 2) * (3 * d->n + d->m) + 64;
}

static void solve_one(const tzo_desc* d, const setup_t* S, const double* xbar0, const double* e0,
                      double* v, double* xbar, double* cost, int32_t* status, int32_t* iters, uint8_t* active, double* wk, int warm, double* rcarry) {
  int nz = S->nz, mi = S->mi, n = d->n, m = d->m, N = d->N, nt = d->ntheta;
  double* th = wk; double* q = th + nt; double* h = q + nz; double* x = h + mi; double* s = x + nz; double* lam = s + mi;
  double* tws = lam + mi; double* iw = tws + (size_t)(d->pmax + 2) * (3 * n + m);
  tube_theta(d, xbar0, e0, th, tws);
  int feas = 1;
  for (int r = 0; r < d->npar; ++r) {
    double a = d->f0[r]; for (int t = 0; t < nt; ++t) a += d->Ft[r * nt + t] * th[t];
    if (!(a >= d->pl[r] - 1e-9) || !(a <= d->pu[r] + 1e-9)) feas = 0;
  }
  int nv = N * m, it = 0, st = 3;
  if (feas) {
    for (int c = 0; c < nz; ++c) { double a = d->q0[c]; for (int t = 0; t < nt; ++t) a += d->Qt[c * nt + t] * th[t]; q[c] = S->c * S->D[c] * a; }
    for (int k = 0; k < mi; ++k) {
      int r = S->row[k]; const double* M = S->sgn[k] > 0 ? d->Ut : d->Lt; double a = S->sgn[k] > 0 ? d->u0[r] : d->l0[r];
      for (int t = 0; t < nt; ++t) a += M[r * nt + t] * th[t];
      h[k] = S->E[k] * S->sgn[k] * a;
    }
    st = ipm(d, S, q, h, x, s, lam, &it, iw, warm, 0.0, (nz > 64 || mi > 1024) ? rcarry : NULL);     /* the device's tile-triangle class */
    if (st != 0) {                      /* same safeguard as the device kernel: once more, cold, textbook fraction to the boundary */
      tzo_desc d2 = *d; int it2 = 0;
      d2.step_frac = fmin(d->step_frac, 0.99);
      st = ipm(&d2, S, q, h, x, s, lam, &it2, iw, 0, 1e-6, NULL);      /* ... and a shifted Newton matrix from the first iteration (a breakdown the
                                                                   * pivot test did not see: tiny positive pivots, garbage step) */
      it += it2;
      if (st != 0 && farkas(S, h, lam)) st = 3;
    }
  }
  if (st != 0) for (int c = 0; c < nz; ++c) x[c] = 0;     /* failed step: v = 0, i.e. u = K e and the nominal state follows Phi (as the device) */
  *status = st; if (iters) *iters = it;
  double obj = 0;
  for (int c = 0; c < nz; ++c) { double px = 0; for (int k = 0; k < nz; ++k) px += S->P[c * nz + k] * x[k]; obj += x[c] * (0.5 * px + q[c]); }
  double r = d->r0;
  for (int i = 0; i < n; ++i) { r += d->r1[i] * xbar0[i]; for (int j = 0; j < n; ++j) r += xbar0[i] * d->R2[i * n + j] * xbar0[j]; }
  *cost = (st == 0) ? obj / S->c + r : INFINITY;
  for (int c = 0; c < nv; ++c) v[c] = S->D[c] * x[c];
  for (int rr = 0; rr < (N + 1) * n; ++rr) {
    double a = 0; for (int j = 0; j < n; ++j) a += d->Phi[rr * n + j] * xbar0[j];
    for (int c = 0; c < nv; ++c) a += d->Gam[rr * nv + c] * v[c];
    xbar[rr] = a;
  }
  if (active) {
    memset(active, 0, (size_t)d->nc);
    if (feas) for (int k = 0; k < mi; ++k) if (s[k] * S->c / (S->E[k] * S->E[k]) < lam[k]) active[S->row[k]] = 1;   /* unscaled slack < unscaled multiplier */
  }
}

int tzo_solve_batch(const tzo_desc* d, int B, const double* xbar0, const double* e0, double* v, double* xbar,
                    double* cost, int32_t* status, int32_t* iters, uint8_t* active, int threads) {
  setup_t* S = make_setup(d);
  size_t wd = work_doubles(d, S);
  int n = d->n, m = d->m, N = d->N;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
    double* wk = (double*)malloc(sizeof(double) * wd);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int b = 0; b < B; ++b)
      solve_one(d, S, xbar0 + (size_t)b * n, e0 + (size_t)b * n, v + (size_t)b * N * m, xbar + (size_t)b * (N + 1) * n,
                cost + b, status + b, iters ? iters + b : NULL, active ? active + (size_t)b * d->nc : NULL, wk, 0, NULL);
    free(wk);
  }
  free_setup(S);
  return 0;
}

/* closed loop, examples/1.double_integrator_sim.py:75-90 */
static int tzo_simulate_core(const tzo_desc* d, int B, int T, const double* x0, const double* noise, const double* At, const double* Bt,
                       double* x_traj, double* u_traj, double* cost, int32_t* status, int threads, int32_t* iters_out);
int tzo_simulate_batch(const tzo_desc* d, int B, int T, const double* x0, const double* noise, const double* At, const double* Bt,
                       double* x_traj, double* u_traj, double* cost, int32_t* status, int threads) {
  return tzo_simulate_core(d, B, T, x0, noise, At, Bt, x_traj, u_traj, cost, status, threads, NULL);
}
/* the same with the interior-point iterations of every step (B x T) reported */
int tzo_simulate_batch_iters(const tzo_desc* d, int B, int T, const double* x0, const double* noise, const double* At, const double* Bt,
                       double* x_traj, double* u_traj, double* cost, int32_t* status, int threads, int32_t* iters_out) {
  return tzo_simulate_core(d, B, T, x0, noise, At, Bt, x_traj, u_traj, cost, status, threads, iters_out);
}
static int tzo_simulate_core(const tzo_desc* d, int B, int T, const double* x0, const double* noise, const double* At, const double* Bt,
                       double* x_traj, double* u_traj, double* cost, int32_t* status, int threads, int32_t* iters_out) {
  setup_t* S = make_setup(d);
  size_t wd = work_doubles(d, S);
  int n = d->n, m = d->m, N = d->N;
  double* wref = NULL;                                   /* work area after the reference solve of the stored start */
  if (d->start_xbar0 && d->warm_floor > 0) {
    wref = (double*)malloc(sizeof(double) * wd);
    double* v0 = (double*)malloc(sizeof(double) * N * m); double* xb0 = (double*)malloc(sizeof(double) * (N + 1) * n);
    double ez[16] = {0}, c0; int32_t st0, it0;
    solve_one(d, S, d->start_xbar0, ez, v0, xb0, &c0, &st0, &it0, NULL, wref, 0, NULL);
    if (st0 != 0) { free(wref); wref = NULL; }              /* reference point not solvable: no stored start (as the device library) */
    free(v0); free(xb0);
  }
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
    double* wk = (double*)malloc(sizeof(double) * wd);
    double* v = (double*)malloc(sizeof(double) * N * m);
    double* xb = (double*)malloc(sizeof(double) * (N + 1) * n);
    double x[16], xbar[16], e[16], u[8], xn[16];
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int b = 0; b < B; ++b) {
      int32_t sticky = 0; int prev_ok = 0; int prev_it = 0; int was_shifted = 0; double rcarry[2] = {0.0, 0.0};
      if (wref) { memcpy(wk, wref, sizeof(double) * wd); prev_ok = 1; was_shifted = -2; }     /* -2: stored start, see below */
      for (int i = 0; i < n; ++i) { x[i] = x0[(size_t)b * n + i]; xbar[i] = x[i]; e[i] = 0; x_traj[((size_t)b * (T + 1)) * n + i] = x[i]; }
      for (int t = 0; t < T; ++t) {
        int32_t st, it; double c;
        int wmode = prev_ok ? ((t & 7) ? 2 : 1) : 0;               /* x / s / lam (and G x) of the previous step live on in wk */
        /* was_shifted: 1 + q = the last start was shifted after q quiet (<= 1 iteration) shifted steps; 0 = not shifted; -1 = not shifted
         * because the quiet budget ran out ("rest"): the unshifted start is kept for as long as it needs no iteration at all (a settled
         * loop: no shift, no G x) and the shifted regime is re-entered the moment it needs one */
        const int seeded = was_shifted == -2;                        /* the stored start is a solution for the start of the loop: taken as it
                                                                      * is, and counted as the first quiet step of the shifted regime */
        const int quiet_run = was_shifted > 0 && prev_it <= 1;
        const int nquiet = quiet_run ? was_shifted : 0;
        const int back = was_shifted == -1 && prev_it >= 1;
        if (seeded) { was_shifted = (d->shift_policy >= 2 && S->srow) ? 1 : 0; wmode = 4; }
        else {
        if (prev_ok && S->srow && (d->shift_policy == 1 || (d->shift_policy >= 2 && (prev_it >= d->shift_policy || back || (quiet_run && (d->shift_quiet == 0 || nquiet <= d->shift_quiet)))))) wmode = 3;
        was_shifted = (wmode == 3) ? 1 + nquiet : ((prev_ok && ((quiet_run && d->shift_policy >= 2) || (was_shifted == -1 && !back))) ? -1 : 0);
        }
        solve_one(d, S, xbar, e, v, xb, &c, &st, &it, NULL, wk, wmode, rcarry);
        prev_it = it;
        if (iters_out) iters_out[(size_t)b * T + t] = it;
        prev_ok = (st == 0) && d->warm_floor > 0;
        if (!sticky && st) sticky = st;
        if (cost) cost[(size_t)b * T + t] = c;
        for (int j = 0; j < m; ++j) { double a = v[j]; for (int i = 0; i < n; ++i) a += d->K[j * n + i] * e[i]; u[j] = a; u_traj[((size_t)b * T + t) * m + j] = a; }
        for (int i = 0; i < n; ++i) {
          double a = noise[((size_t)b * T + t) * n + i];
          for (int j = 0; j < n; ++j) a += At[i * n + j] * x[j];
          for (int j = 0; j < m; ++j) a += Bt[i * m + j] * u[j];
          xn[i] = a;
        }
        for (int i = 0; i < n; ++i) { x[i] = xn[i]; xbar[i] = xb[n + i]; e[i] = x[i] - xbar[i]; x_traj[((size_t)b * (T + 1) + t + 1) * n + i] = x[i]; }
      }
      status[b] = sticky;
    }
    free(wk); free(v); free(xb);
  }
  free(wref);
  free_setup(S);
  return 0;
}

int tzo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
