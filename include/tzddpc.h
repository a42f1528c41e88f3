/*
 * tzddpc.h -- C-ABI of the MI355X (gfx950) TZDDPC hot path.
 *
 * The reference (rssalessio/TZDDPC) is pure Python and has no FFI; its boundary for this path is the
 * Python method surface of class TZDDPC.  Each entry point below names the reference interface it
 * stands behind (paths relative to the reference repository).  The Python binding that calls these
 * through ctypes is tzddpc_amd/native.py; INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - all matrices row-major, double precision, caller-owned; the library copies what it keeps and
 *     never frees caller memory;
 *   - every function returns 0 on success, a negative tz_status on failure; tz_last_error() gives the
 *     message of the last failure on the calling thread;
 *   - one tz_problem belongs to one device; calls on one tz_problem must be serialised by the caller
 *     (the reference object is not thread-safe either: all state hangs off the TZDDPC instance);
 *   - batch pointers (xbar0, e0, v, xbar, ...) are HOST pointers when mem == TZ_MEM_HOST and DEVICE
 *     pointers (e.g. torch tensor .data_ptr()) when mem == TZ_MEM_DEVICE; with device pointers a call
 *     only enqueues work on the problem's stream (tz_problem_sync / tz_problem_stream to order it).
 */
#ifndef TZDDPC_H
#define TZDDPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TZ_ABI_VERSION 4

typedef enum tz_status {
  TZ_OK = 0,
  TZ_ERR_INVALID = -1,   /* bad argument / inconsistent sizes            */
  TZ_ERR_HIP = -2,       /* HIP runtime error (no device, launch failure) */
  TZ_ERR_UNSUPPORTED = -3
} tz_status;

/* per-trajectory solve status (int32 array `status`) */
enum {
  TZ_SOLVED = 0,
  TZ_MAX_ITER = 1,          /* iteration cap hit before the tolerances                              */
  TZ_NUMERICAL = 2,         /* non-finite iterate, a factorisation that fails even with the largest */
                            /* diagonal shift, or complementarity collapsed before the residuals    */
  TZ_INFEASIBLE = 3,        /* a parameter-only constraint is violated (e.g. xbar0 + e0 outside X,  */
                            /* reference tzddpc/tzddpc.py:191-195 at k = 0), or a failed solve       */
                            /* whose multipliers are a Farkas certificate of primal infeasibility   */
                            /* (y = lam / max lam >= 0, |G'y| <= 1e-6, h'y < -1e-6); the reference   */
                            /* raises Exception('Problem is unbounded') for an infeasible problem    */
                            /* (tzddpc/tzddpc.py:374-375)                                            */
};

enum { TZ_MEM_HOST = 0, TZ_MEM_DEVICE = 1 };

/* Sparse affine map  out[i] = c0[i] + sum_{e in [ptr[i], ptr[i+1])} val[e] * theta[col[e]]  (CSR). */
typedef struct tz_affmap {
  int32_t rows;
  const int32_t* ptr;   /* rows + 1 */
  const int32_t* col;
  const double* val;
  const double* c0;     /* rows */
} tz_affmap;

/*
 * Everything TZDDPC.build_problem / build_problem_simplified fixes at build time
 * (reference tzddpc/tzddpc.py:132-241, :243-355), already collapsed and equilibrated by the host
 * (tzddpc_amd/builder.py, tzddpc_amd/tzddpc.py):
 *
 *   per trajectory and MPC step the device solves, for parameters (xbar0, e0),
 *       minimise 1/2 x'P x + q(theta)'x     subject to   G x <= h(theta)
 *   theta = [ xbar0 | |xbar0| | (c_k, rho^x_k, rho^u_k)_{k<N} ]  with the last block computed on the
 *   device from e0 by the tube recursion (the numeric content of reference :172-181, :191-192).
 */
typedef struct tz_problem_desc {
  int32_t abi_version;      /* TZ_ABI_VERSION */
  int32_t n, m, N;          /* dim_x, dim_u, horizon (reference :30-43, :134) */
  int32_t nz, mi;           /* decision variables, one-sided inequality rows */
  int32_t ntheta;           /* 2 n + N (2 n + m) */
  const double* P;          /* nz x nz, symmetric PSD, scaled */
  const double* G;          /* mi x nz, scaled */
  tz_affmap q;              /* nz rows  (already times c D)  */
  tz_affmap h;              /* mi rows  (already times E)    */
  tz_affmap par;            /* parameter-only rows, feasible iff par_lo <= value <= par_hi */
  const double* par_lo;
  const double* par_hi;
  /* objective:  (1/2 x'Px + q'x) / cost_scale + r0 + r1'xbar0 + xbar0'R2 xbar0  == the value
   * cvxpy's problem.solve() returns (reference :367)                                            */
  double cost_scale;
  double r0;
  const double* r1;         /* n */
  const double* R2;         /* n x n */
  const double* Dz;         /* nz: z = Dz .* x  (undo column equilibration) */
  const double* Phi;        /* (N+1) n x n     xbar = Phi xbar0 + Gam v   (reference :166-170) */
  const double* Gam;        /* (N+1) n x N m */
  /* multipliers per row of the two-sided problem the host built (for the active-set report) */
  int32_t nc_rows;          /* rows of that problem */
  const int32_t* row_of;    /* mi: which of those rows an inequality row belongs to */
  const double* act_scale;  /* mi: c / E_r^2; row r is reported active iff s_r * act_scale_r < lambda_r, i.e. iff the
                             * UNSCALED slack is smaller than the UNSCALED multiplier (strict-complementarity partition) */
  /* tube constants (reference :119-128 after reduce(1); :175, :181) */
  const double* CK;         /* n x n  center of MdataK = Ahat + Bhat K */
  const double* DK;         /* n x n  sum_i |G_i| of MdataK's single-entry generators */
  const double* K;          /* m x n  theta.K */
  int32_t pmax;             /* highest power of M_K applied to <e0, 0> */
  const double* absCKpow;   /* pmax x n x n   |C_K^j| */
  const double* absKCKpow;  /* pmax x m x n   |K C_K^j| */
  const int32_t* power;     /* N: the e0 part of Ze_k is M_K^power[k] <e0, 0> (full: k; simplified :292-295) */
  /* interior-point options */
  int32_t max_iter;
  double tol;               /* scaled residual / complementarity tolerance */
  double reg;               /* static diagonal regularisation of the reduced Newton matrix */
  double step_frac;         /* fraction of the step to the boundary */
  /* receding-horizon shift of the warm start of closed-loop steps (all four may be NULL: never shift).  Variable c starts from
   * x_prev[shift_var[c]] * shift_xscale[c], the multiplier of row r from lambda_prev[shift_row[r]] * shift_lscale[r] (the scales
   * undo / redo the equilibration of the source and target).  Whether a step shifts is the policy of tz_problem_set_warm_shift. */
  const int32_t* shift_var;     /* nz */
  const int32_t* shift_row;     /* mi */
  const double* shift_xscale;   /* nz */
  const double* shift_lscale;   /* mi */
  /* recovery of v when the host has eliminated equality constraints of build_constraints (reference tzddpc/tzddpc.py:213-219
   * accepts any DCP constraint, `==` included): the first N m variables are then no longer v itself and
   *     v = rec_c0 + rec_x0 xbar0 + rec_y x          (N m; N m x n; N m x nz, x the SCALED variables: Dz is folded in)
   * replaces v = Dz .* x[:N m].  All three NULL: no elimination. */
  const double* rec_c0;
  const double* rec_x0;
  const double* rec_y;
  /* Plan overrides (0 = the library chooses).  The library picks its code paths from the problem's size and structure; these bits
   * force the general-size paths on a problem that would qualify for a specialised one -- the parity tests use them to hold the
   * paths against each other on the same problem.  The release library reads NO environment variables. */
  int32_t plan_flags;       /* TZ_PLAN_* */
} tz_problem_desc;

enum {
  TZ_PLAN_UNFUSED = 1,            /* closed-loop entry points run tube / affine / solve / finish / plant as separate launches */
  TZ_PLAN_GENERAL_CHOLESKY = 2,   /* nz <= 64: four-wave factorisation and LDS-published solves instead of the one-wave forms */
  TZ_PLAN_ITEM_GRAM = 4,          /* nz <= 40: item-plan Gram instead of the super-step form */
  TZ_PLAN_NO_STAIRCASE = 8,       /* nz > 64: keep the caller's variable / row order */
  TZ_PLAN_ELL_PRODUCTS = 16       /* G x / G'v by the lane-ELL walks even when G is block-Toeplitz over the horizon */
};

typedef struct tz_problem tz_problem;

/* ABI / device discovery */
int tz_abi_version(void);
const char* tz_last_error(void);
int tz_device_count(int* count);

/*
 * K0 -- identification of B data sets on the device (Monte Carlo over data seeds): what TZDDPC.build_zonotopes +
 * build_zonotopes_theta compute per data set (reference tzddpc/tzddpc.py:67-85, :119-128), in the closed form the Girard order-1
 * reduce(1) of :126-128 gives (valid when W has more than n (n + m) / (T - 1) generators, i.e. always for the examples' T):
 *   u : B x T x m, x : B x T x n          input / state data (reference :20-28; row t of x is the state BEFORE input row t acts)
 *   w_center : n                          centre of W
 *   K : m x n (k_shared != 0) or B x m x n, may be NULL
 *   C  : B x n x (n+m)  (out)             centre of Mdata = [A_hat | B_hat]                                   (:83, :163)
 *   s  : B x (n+m)      (out)             sum_t |pinv([Xm'; Um'])[t, :]|: the boxed generators of Mdata / Mdelta are rad(W) s'
 *   sK : B x n          (out, NULL if K is)   the same for MdataK = Mdata [I; K]: rad(W) sK'                   (:119)
 *   CK : B x n x n      (out, NULL if K is)   centre of MdataK = A_hat + B_hat K
 *   status : B int32    (out)             0, or TZ_NUMERICAL when the data are not persistently exciting (Gram matrix singular)
 * Gram contractions on v_mfma_f64_4x4x4, one wave per data set.
 */
int tz_identify_batch(int device, int32_t B, int32_t T, int32_t n, int32_t m, const double* u, const double* x,
                      const double* w_center, const double* K, int32_t k_shared,
                      double* C, double* s, double* sK, double* CK, int32_t* status, int mem);

/*
 * Gain synthesis without MOSEK (reference tzddpc/utils.py).  Both calls evaluate closed-loop matrices
 *     M(beta) = M0 + sum_i beta_i H_i        (n x n, n <= 8; H : ngen x n x n row-major; host pointers; blocking)
 * for S coefficient vectors at once, one GPU lane per vector.
 *
 * tz_specrad_batch -- the sampling test of is_gain_robust (reference tzddpc/utils.py:105-129): with M0 = A0 + B0 K,
 *   H_i = G_i[:, :n] + G_i[:, n:] K and beta = S samples of Mdata.sample()'s coefficients, rho[s] = spectral_radius(A + B K)
 *   (:8-11; in-LDS Hessenberg reduction + Francis double-shift QR per lane), status[s] = 0 or TZ_NUMERICAL (QR did not converge).
 *
 * tz_adversary_batch -- the adversarial model search of compute_A_B (reference tzddpc/utils.py:13-41): maximise ||A + B K||_F
 *   over independent beta_A, beta_B in [-1, 1]^gamma, H = [G_i[:, :n]]_i followed by [G_i[:, n:] K]_i (ngen = 2 gamma).  From
 *   every starting point beta0[s] the convex-concave procedure the reference delegates to DCCP (:37) is iterated to its fixed
 *   point: one step sets all beta_i = sign <M(beta), H_i> (the vertex of the box that maximises the linearised objective; a zero
 *   inner product keeps beta_i).  beta[s] (out) = the fixed point, fro[s] = ||M(beta[s])||_F, steps[s] = CCP steps taken
 *   (<= max_iter).  The caller keeps the best start (the reference's ccp_times = num_init).
 */
int tz_specrad_batch(int device, int32_t S, int32_t n, int32_t ngen, const double* M0, const double* H, const double* beta,
                     double* rho, int32_t* status);
int tz_adversary_batch(int device, int32_t S, int32_t n, int32_t ngen, const double* M0, const double* H, const double* beta0,
                       int32_t max_iter, double* beta, double* fro, int32_t* steps);

/*
 * K1g -- literal stacked-generator tubes (the general path: any generators of MdataK / Mdelta, boxed or dense).
 * Every generator column of every tube Ze[k] the reference builds by zonotope algebra (tzddpc/tzddpc.py:172-207, :283-324) is
 * g = m0 + M xi_src with xi_src = e0 (src 0), zeta_j = [xbar_j; v_j] (src 1 + j) or nothing (src -1); the host lists them in the
 * reference's order (tzddpc_amd/genstack.py), the device streams the list from HBM and evaluates per trajectory
 *     centre_k,  rad^x_k = sum_g |g|,  rad^u_k = sum_g |K g|        -- Ze[k].interval, (Ze[k] * K).interval of :191-192
 * (tz_genstack_intervals) or the columns [centre | generators] of one tube, the Z.value of the CVXZonotope solve() returns at
 * :377 (tz_genstack_values).
 */
typedef struct tz_genstack_desc {
  int32_t n, m, N, nseg;          /* dim_x, dim_u, horizon, number of tubes Ze[0 .. nseg-1] */
  const int32_t* seg_ptr;         /* nseg + 1: generators of Ze[k] are [seg_ptr[k], seg_ptr[k+1]) */
  const int32_t* src;             /* G */
  const double* m0;               /* G x n */
  const double* M;                /* G x n x (n+m), columns beyond the width of the source zero */
  const double* c0;               /* nseg x n            centre_k = c0 + cE e0 + sum_j cZ[k][j] zeta_j */
  const double* cE;               /* nseg x n x n */
  const double* cZ;               /* nseg x N x n x (n+m), may be NULL (zero: Mdelta has a zero centre, :122-123) */
  const double* K;                /* m x n */
} tz_genstack_desc;
typedef struct tz_genstack tz_genstack;
int tz_genstack_create(int device, const tz_genstack_desc* desc, tz_genstack** out);
int tz_genstack_destroy(tz_genstack* g);
/* e0 : B x n, zeta : B x N x (n+m);  centre, rad_x : B x nseg x n, rad_u : B x nseg x m;  kernel_ms (may be NULL): HIP-event
 * time of the streaming kernel of this call */
int tz_genstack_intervals(tz_genstack* g, int32_t B, const double* e0, const double* zeta,
                          double* centre, double* rad_x, double* rad_u, double* kernel_ms, int mem);
/* Literal problems (matrix zonotopes with dense generators: one epigraph variable per decision-dependent generator entry, built by
 * tzddpc_amd/builder.py with literal=...): the decision-INDEPENDENT generators of every tube (constants and M_K^p <e0, 0>, reference
 * tzddpc/tzddpc.py:175, :181) stay numeric; theta's tube block (centre, rho^x, rho^u per step) is then the interval hull of that
 * sub-stack, evaluated per solve by the K1g kernels on the problem's stream instead of the collapsed recursion.  The stack is not
 * owned (destroy it after the problem); closed-loop entry points of such a problem run the four-kernel step (no fused launch).
 * NULL detaches. */
int tz_problem_attach_tube_stack(tz_problem* p, tz_genstack* stack);

/* Z : B x n x (1 + Gamma_seg), column 0 the centre, then the generators of Ze[seg] in the reference's order */
int tz_genstack_values(tz_genstack* g, int32_t seg, int32_t B, const double* e0, const double* zeta, double* Z, int mem);
/* generators, bytes of the stack streamed per tile of 256 trajectories, chunks (= workgroups per tile) */
int tz_genstack_info(tz_genstack* g, int64_t* generators, int64_t* stack_bytes, int64_t* chunks);

/* Build-time: upload one problem to `device`.  Stands behind the *result* of
 * TZDDPC.build_problem / build_problem_simplified (reference tzddpc/tzddpc.py:132, :243), i.e. the
 * object `self.problem_full` that solve() later consumes.
 * Sizes: dim_x <= 16, dim_u <= 8, nz <= 256 decision variables, mi <= 1536 one-sided rows, pmax <= 128 (TZ_ERR_UNSUPPORTED beyond;
 * the reference itself takes any size, tzddpc/tzddpc.py:155-160).  tz_identify_batch: dim_x <= 8, dim_u <= 4; tz_specrad_batch /
 * tz_adversary_batch: n <= 8. */
int tz_problem_create(int device, const tz_problem_desc* desc, tz_problem** out);
int tz_problem_destroy(tz_problem* p);

/* Stream control (plumbing for torch interop): `stream` is a hipStream_t, NULL = library-owned stream. */
int tz_problem_set_stream(tz_problem* p, void* stream);
int tz_problem_sync(tz_problem* p);

/*
 * TZDDPC.solve(xbar0, e0, **kw) for a batch of B independent trajectories
 * (reference tzddpc/tzddpc.py:357-377; the loop bodies examples/1.double_integrator_sim.py:76-80).
 *   xbar0, e0 : B x n           (in)
 *   v         : B x N x m       (out)   == self.variables[0].value
 *   xbar      : B x (N+1) x n   (out)   == self.variables[1].value
 *   cost      : B               (out)   == result
 *   status    : B int32         (out)   see enum above; the Python layer maps != TZ_SOLVED of a
 *                                       single-trajectory call to the reference's exceptions
 *   iters     : B int32         (out, may be NULL)
 *   active    : B x nc_rows uint8 (out, may be NULL)  1 = constraint row active at the optimum
 */
int tz_solve_batch(tz_problem* p, int32_t B, const double* xbar0, const double* e0,
                   double* v, double* xbar, double* cost, int32_t* status, int32_t* iters,
                   uint8_t* active, int mem);

/*
 * Closed loop over T steps for B trajectories: the loop of examples/1.double_integrator_sim.py:75-90
 * (solve -> xbar+ = xbar[1] -> u = K e + v[0] -> x+ = A x + B u + w -> e+ = x+ - xbar+), one launch
 * sequence per step, no host synchronisation inside.
 *   x0      : B x n                    initial state (xbar starts at x0, e at 0, :68-70)
 *   noise   : B x T x n                process noise realisations w_t
 *   A_true  : n x n, B_true : n x m    plant
 *   x_traj  : B x (T+1) x n  (out)     u_traj : B x T x m (out)
 *   cost    : B x T (out, may be NULL) status : B int32 (out) first non-zero step status, sticky
 */
int tz_simulate_batch(tz_problem* p, int32_t B, int32_t T, const double* x0, const double* noise,
                      const double* A_true, const double* B_true,
                      double* x_traj, double* u_traj, double* cost, int32_t* status, int mem);

/* One closed-loop step on device-resident state (what bench.py times): state buffers are device
 * pointers owned by the caller: x, xbar, e : B x n (in/out); w : B x n (in); u_out : B x m. */
int tz_mpc_step(tz_problem* p, int32_t B, double* x, double* xbar, double* e, const double* w,
                const double* A_true, const double* B_true, double* u_out, double* cost, int32_t* status);

/* K consecutive closed-loop steps of every trajectory in ONE kernel launch (each workgroup loops over the steps of its
 * trajectory; state and warm start stay on chip): w holds the noise of the K steps, step-major (K x B x n); u_out / cost keep
 * the values of the last step; status is the sticky first non-zero status.  Same results as K calls of tz_mpc_step. */
int tz_mpc_run(tz_problem* p, int32_t B, int32_t K, double* x, double* xbar, double* e, const double* w,
               const double* A_true, const double* B_true, double* u_out, double* cost, int32_t* status);

/* Stored start.  Solves the problem once, cold, for the parameters (xbar0, e0) (host pointers, n doubles each; typically the centre of
 * X0 and e0 = 0: where every closed loop of the reference's examples begins, examples/1.double_integrator_sim.py:62-70) and keeps
 * the solution and its multipliers with the handle.  From then on the first step of a closed loop that has no previous solution
 * (a fresh tz_simulate_batch, the first tz_mpc_step / tz_mpc_run after tz_problem_reset_warm) starts every trajectory from that
 * point -- a warm start like any other: same optimum, fewer iterations (13 -> 0 ... 5 for the double integrator N = 20) -- instead of
 * from the cold point.  tz_solve_batch is not affected (always cold).  NULL pointers forget the stored start.  Fails (nothing
 * stored) when the reference point itself is not solvable. */
int tz_problem_store_start(tz_problem* p, const double* xbar0, const double* e0);

/* Warm-start shift policy of the closed-loop entry points: 0 never (default), 1 every warm-started step, k >= 2 only the steps
 * that follow a step of at least k interior-point iterations (the transient) and, after those, for as long as the shifted steps
 * finish in one iteration (up to tz_problem_set_warm_quiet of them).  Which one pays depends on the problem (double
 * integrator N=20: 1; pulley: 0) -- the Python layer calibrates it on a short simulated closed loop at build time. */
int tz_problem_set_warm_shift(tz_problem* p, int32_t policy);
/* Policies k >= 2 leave the shifted regime after `quiet_steps` shifted steps in a row that needed at most one iteration (default 16,
 * 0: never) and REST: the previous solution is taken as it is (no shift, G x of the previous iterate kept: the cheapest step there
 * is) for as long as such a start needs no iteration at all; the first resting step that needs one sends the trajectory back to the
 * shifted regime.  (Until round 4 a trajectory came back only after a step of >= k iterations, and a loop that had not settled
 * when its budget ran out took one Newton step in every step from then on.) */
int tz_problem_set_warm_quiet(tz_problem* p, int32_t quiet_steps);

/* Stopping test of the interior point, relative to the `tol` of the descriptor: scaled residuals <= res_factor * tol and
 * complementarity mu <= mu_factor * tol (defaults 100 and 1e-3: 1e-8 / 1e-13 at tol = 1e-10).  The distance to the solution of a
 * degenerate problem goes like sqrt(mu), so mu_factor is what carries the accuracy of the closed loop; how small it has to be
 * depends on the problem (double integrator N = 80: 1e-3 for 2e-8, LP-type losses: 0.3 is already at 3e-9) -- the Python layer
 * picks the loosest factor that keeps a simulated closed loop within 2e-8 of the tightest one (TZDDPC.mu_factor). */
int tz_problem_set_stopping(tz_problem* p, double res_factor, double mu_factor);

/* Warm-started steps re-derive the slacks for the new right-hand side and push the point into the cone: with
 *     sigma = min( max(floor, gain * (largest violation of the new rows by the previous solution)), cap )      (scaled units)
 * slack >= sigma and multiplier >= sigma^2 / slack (onto the central path of mu = sigma^2; inactive rows keep multipliers ~ 0).
 * The rows are equilibrated, so a slack of order one is already far inside: violations of 5 ... 70 (the first steps of a
 * transient, any step on a plant that differs from the model) would otherwise wipe out the slack information of the start; with
 * the cap the best settings no longer depend on the plant (measured in the C oracle on the example's true plant and on the
 * identified centre).  Gain and cap that cost the fewest interior-point iterations depend on the problem -- the Python layer
 * calibrates them together with the shift policy.  Defaults: floor 1e-8, gain 1, no cap. */
int tz_problem_set_warm_push(tz_problem* p, double floor, double gain, double cap);

/* The closed-loop entry points (tz_mpc_step, tz_mpc_run) warm-start every trajectory from the solution the HANDLE holds for it from
 * the previous call with the same batch size.  Call this when the next call starts a new batch of trajectories (fresh x / xbar / e):
 * the next step then starts cold, and results and iteration counts no longer depend on what the handle solved before.
 * (tz_solve_batch and tz_simulate_batch never use the stored state.) */
int tz_problem_reset_warm(tz_problem* p);

/* Kernel timing with HIP events on the problem's stream (bench.py roofline leg).
 * kernel ids: 0 = tz_prepare, 1 = tz_ipm, 2 = tz_finish, 3 = tz_plant_step */
int tz_timing_enable(tz_problem* p, int enable);
int tz_timing_get(tz_problem* p, int kernel, double* total_ms, int64_t* launches);
/* Work done by tz_ipm since tz_timing_enable(p, 1): number of Newton-matrix factorisations (Gram + Cholesky) summed over all
 * trajectories and launches, number of trajectory solves, and (may be NULL) the largest number of factorisations any ONE trajectory
 * did inside one launch -- with all trajectories resident at once a multi-step launch lasts as long as its slowest trajectory.
 * Counted on the device by the kernel itself. */
int tz_ipm_work_get(tz_problem* p, int64_t* factorizations, int64_t* trajectory_solves, int64_t* max_factorizations_one_trajectory);

/* Static plan of tz_ipm for one trajectory and one interior-point iteration: number of
 * v_mfma_f64_4x4x4 instructions that carry useful tiles in the Gram formation (G'WG, block-sparse) and in
 * the Cholesky trailing updates, the number actually issued (padding included), LDS bytes per workgroup
 * and bytes of the packed constraint patches. */
int tz_ipm_plan_info(tz_problem* p, int64_t* mfma_gram_per_iter, int64_t* mfma_chol_per_iter,
                     int64_t* mfma_issued_per_iter, int64_t* lds_bytes, int64_t* patch_bytes);

/* The code paths the library chose for this problem (each 0 / 1): one-launch closed-loop step; one-wave factorisation with the
 * trailing forward substitution (nz <= 64); super-step Gram (nz <= 40); staircase ordering (nz > 64); G x / G'v as block-Toeplitz
 * convolutions over the horizon.  See tz_problem_desc.plan_flags. */
int tz_problem_plan_get(tz_problem* p, int32_t* fused, int32_t* one_wave_cholesky, int32_t* superstep_gram, int32_t* staircase,
                        int32_t* toeplitz);

/* Test hook: copy device-side intermediates of trajectory `b` of the last solve to the host.
 * what: 0 = theta (ntheta), 1 = q (nz), 2 = h (mi)  [0-2: only after tz_solve_batch, the closed-loop launches keep them on chip],
 *       3 = x (nz), 4 = s (mi), 5 = lambda (mi), 6 = per-phase cycle sums (diagnostic build), 7 = iterations of every trajectory */
int tz_debug_fetch(tz_problem* p, int32_t b, int what, double* out, int32_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* TZDDPC_H */
