"""Shared test helpers: the reference examples' callbacks written against cplite, system construction."""
import numpy as np

from tzddpc_amd import cplite as cp


def loss_di(u, x):            # reference examples/1.double_integrator_sim.py:22-28
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(x[i, :], p=2) ** 2 + 1e-2 * cp.norm(u[i], p=1)
    return cost


def loss_pulley(u, y):        # reference examples/2.pulley_sim.py:17-22
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(y[i, 0] - 1, p=2)
    return cost


def loss_dim5(u, x):          # reference examples/3.5dimsystem_sim.py:14-20
    cost = 0
    for i in range(u.shape[0]):
        cost += 1e9 * cp.norm(x[i, 1] - 2, p=2) + 1e-1 * cp.norm(u[i], p=2)
    return cost


def loss_dim5q(u, x):         # strictly convex loss on the 5-dim systems (oracle.harness.loss_dim5_quadratic): determines v with two inputs
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(x[i, :] - np.array([0.0, 3.0, 0.0, 0.0, 0.0]), p=2) ** 2 + 1e-2 * cp.norm(u[i], p=1)
    return cost


def cons_dim5(u, x):          # reference examples/3.5dimsystem_sim.py:23-26
    return [x[:, 1] <= 10, x[:, 1] >= 2]


def nocons(u, x):
    return []


def cons_di_eq(v, x):         # `==` rows (reference tzddpc/tzddpc.py:213-219 takes any DCP constraint): terminal state, move blocking
    return [x[x.shape[0] - 1, :] == np.array([-4.0, 0.0]), v[3] == v[4], v[5, 0] <= 0.9]


# cases whose optimum is a face, not a point (the loss prices one state coordinate only and there are two inputs): parity on the
# objective and the priced coordinate; input / state trajectories only against a solver on the SAME formulation (analytic centre)
NONUNIQUE = {"dim5m2_n20"}

CASES = {
    # name: (system, loss, constraints, horizon, k0)
    "di_n2": ("di_sim", loss_di, nocons, 2, None),
    "di_sim_n5": ("di_sim", loss_di, nocons, 5, None),
    "di_n20_k2": ("di_cc", loss_di, nocons, 20, 2),
    "di_n80": ("di_cc", loss_di, nocons, 80, None),
    "di_n5": ("di_cc", loss_di, nocons, 5, None),
    "di_n20": ("di_cc", loss_di, nocons, 20, None),
    "di_n20_k1": ("di_cc", loss_di, nocons, 20, 1),
    "di_n10": ("di_cc", loss_di, nocons, 10, None),
    "di_n40": ("di_cc", loss_di, nocons, 40, None),
    "di_n10_eq": ("di_cc", loss_di, cons_di_eq, 10, None),
    "pulley_n10": ("pulley", loss_pulley, nocons, 10, None),
    "dim5_n20": ("dim5_w001", loss_dim5, cons_dim5, 20, None),
    # m = 2: the 5-dim system as BASELINE.json configs[3] states it (the loss's ||u_i||_2 is on the free variable of build_problem
    # and vanishes, reference tzddpc/tzddpc.py:160,222), and a two-input double integrator
    "dim5m2_n20": ("dim5m2_w001", loss_dim5, cons_dim5, 20, None),
    "dim5m2q_n20": ("dim5m2_w001", loss_dim5q, cons_dim5, 20, None),
    "di2in_n10": ("di2in", loss_di, nocons, 10, None),
    "di2in_n10_k1": ("di2in", loss_di, nocons, 10, 1),
}


def identified_qp(case, seed=25, epigraph="auto", loss=None):
    """Host-only: data -> Mdata -> collapsed parametric QP, without touching the GPU (`loss` replaces the case's callback)."""
    from tzddpc_amd import TZDDPC
    from tzddpc_amd.builder import build_parametric_qp
    from tzddpc_amd.harness import generate_trajectories, system
    sysname, loss_case, cons, N, k0 = CASES[case]
    loss = loss or loss_case
    A, B, zon, T = system(sysname)
    rng = np.random.default_rng(seed)
    ctl = TZDDPC.__new__(TZDDPC)
    ctl.device = 0; ctl._native = None; ctl.qp = None
    ctl.update_identification_data(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, rng))
    ctl.build_zonotopes_theta(zon)
    n = ctl.dim_x
    Xi, Ui = zon.X.interval, zon.U.interval
    qp = build_parametric_qp(ctl.Mdata.center[:, :n], ctl.Mdata.center[:, n:], ctl.MdataK.center,
                             ctl.MdataK.single_entry_magnitudes(), ctl.Mdelta.single_entry_magnitudes(), ctl.theta.K,
                             zon.W.center, zon.W.generators, Xi.left_limit, Xi.right_limit, Ui.left_limit, Ui.right_limit,
                             N, loss, cons, k0, epigraph=epigraph)
    return ctl, qp, (A, B, zon)


def gpu_controller(case, seed=25, **solver_kwargs):
    from tzddpc_amd import TZDDPC
    from tzddpc_amd.harness import generate_trajectories, system
    sysname, loss, cons, N, k0 = CASES[case]
    A, B, zon, T = system(sysname)
    rng = np.random.default_rng(seed)
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, rng))
    ctl.build_zonotopes_theta(zon)
    if k0 is None:
        ctl.build_problem(N, loss, cons, **solver_kwargs)
    else:
        ctl.build_problem_simplified(k0, N, loss, cons, **solver_kwargs)
    return ctl, (A, B, zon)


def sample_params(zon, n, B, seed=7, e_scale=0.02, x_scale=0.05):
    rng = np.random.default_rng(seed)
    x0 = np.tile(zon.X0.center, (B, 1)) + x_scale * rng.standard_normal((B, n))
    e0 = e_scale * rng.standard_normal((B, n))
    e0[0] = 0.0
    return x0, e0


def oracle_solution(qp, x0, e0, tol=1e-12):
    """High-accuracy reference for one (xbar0, e0): numpy interior point + KKT certificate."""
    from oracle.qp_ipm import solve_qp
    from tzddpc_amd.builder import theta_reference
    th = theta_reference(qp, x0, e0)
    ql = qp.q0 + qp.Qt @ th; ll = qp.l0 + qp.Lt @ th; ul = qp.u0 + qp.Ut @ th
    r = solve_qp(qp.P, ql, qp.A, ll, ul, tol=tol)
    nv = qp.N * qp.m
    v = r.x[:nv].reshape(qp.N, qp.m)
    xbar = (qp.Phi @ x0 + qp.Gam @ r.x[:nv]).reshape(qp.N + 1, qp.n)
    cost = r.obj + qp.r0 + qp.r1 @ x0 + x0 @ qp.R2 @ x0
    slack = np.minimum(np.where(np.isfinite(ul), ul - qp.A @ r.x, np.inf), np.where(np.isfinite(ll), qp.A @ r.x - ll, np.inf))
    active = slack < np.abs(r.y)
    return dict(v=v, xbar=xbar, cost=cost, status=r.status, cert=r.cert, active=active, y=r.y, slack=slack)


def golden_tube_rows(qp):
    """Product row index of every tube row, keyed like the golden arrays: {(k, component, side): row} with component < n the
    state rows and n + j the input rows, side 0 upper / 1 lower.  Rows of step 0 are parameter-only in the product (absent)."""
    import re
    out = {}
    pat = re.compile(r"^(X|U)(ub|lb)\[(\d+),(\d+)\]$")
    for r, name in enumerate(qp.row_names):
        mt = pat.match(name)
        if mt:
            comp = int(mt.group(4)) + (qp.n if mt.group(1) == "U" else 0)
            out[(int(mt.group(3)), comp, 0 if mt.group(2) == "ub" else 1)] = r
    return out


def host_controller_from_golden(case, g):
    """Host-only product controller (no GPU) identified from a golden file's data set and gain."""
    from tzddpc_amd import TZDDPC, Data, Theta
    from tzddpc_amd.harness import system
    sysname, loss, cons, N, k0 = CASES[case]
    A, B, zon, T = system(sysname)
    ctl = TZDDPC.__new__(TZDDPC)
    ctl.device = 0; ctl._native = None; ctl.qp = None
    ctl.update_identification_data(Data(g["data_u"], g["data_x"]))
    ctl.build_zonotopes_theta(zon, theta=Theta(g["K"], np.zeros_like(A), np.zeros_like(B)))
    return ctl, (A, B, zon)


def qp_from_golden(case, g, epigraph="auto"):
    from tzddpc_amd.builder import build_parametric_qp
    sysname, loss, cons, N, k0 = CASES[case]
    ctl, (A, B, zon) = host_controller_from_golden(case, g)
    n = ctl.dim_x
    Xi, Ui = zon.X.interval, zon.U.interval
    return build_parametric_qp(ctl.Mdata.center[:, :n], ctl.Mdata.center[:, n:], ctl.MdataK.center,
                               ctl.MdataK.single_entry_magnitudes(), ctl.Mdelta.single_entry_magnitudes(), ctl.theta.K,
                               zon.W.center, zon.W.generators, Xi.left_limit, Xi.right_limit, Ui.left_limit, Ui.right_limit,
                               N, loss, cons, k0, epigraph=epigraph)


def c_oracle_for(ctl, **kw):
    """Plain-C oracle on the controller's QP with the warm-start policy the device chose (the shift maps are data handed to it)."""
    from oracle.c_oracle import COracle
    from tzddpc_amd.builder import horizon_shift
    pol = kw.pop("shift_policy", ctl.warm_shift_policy)
    kw.setdefault("warm_gain", float(getattr(ctl, "warm_push_gain", 1.0)))
    kw.setdefault("mu_factor", float(getattr(ctl, "mu_factor", 1e-3)))
    kw.setdefault("warm_cap", float(getattr(ctl, "warm_push_cap", 1e300)))
    ss = getattr(ctl, "stored_start", None)                       # fresh closed loops begin from the stored start on both sides
    kw.setdefault("stored_start", None if ss is None else ss[0])
    return COracle(ctl.qp, shift_policy=pol, shift_maps=horizon_shift(ctl.qp) if pol else None, **kw)
