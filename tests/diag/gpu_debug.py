"""First-contact GPU check: device theta / q / h vs host formulas, solve vs oracle IPM."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from tzddpc_amd import TZDDPC, cplite as cp
from tzddpc_amd.harness import system, generate_trajectories
from tzddpc_amd.builder import theta_reference
from oracle.qp_ipm import solve_qp

def loss_di(u, x):
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(x[i, :], p=2) ** 2 + 1e-2 * cp.norm(u[i], p=1)
    return cost
def loss_pulley(u, y):
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(y[i, 0] - 1, p=2)
    return cost
def loss_dim5(u, x):
    cost = 0
    for i in range(u.shape[0]):
        cost += 1e9 * cp.norm(x[i, 1] - 2, p=2) + 1e-1 * cp.norm(u[i], p=2)
    return cost
cons_dim5 = lambda u, x: [x[:, 1] <= 10, x[:, 1] >= 2]
nocons = lambda u, x: []
cases = [("di_cc", loss_di, nocons, 5, None), ("di_cc", loss_di, nocons, 20, None), ("di_cc", loss_di, nocons, 20, 1),
         ("pulley", loss_pulley, nocons, 10, None), ("dim5_w001", loss_dim5, cons_dim5, 20, None)]
only = sys.argv[1:] 
for name, loss, cons, N, k0 in cases:
    A, B, zon, T = system(name)
    rng = np.random.default_rng(25)
    data = generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, rng)
    c = TZDDPC(data)
    c.build_zonotopes_theta(zon)
    t0 = time.time()
    if k0 is None: c.build_problem(N, loss, cons)
    else: c.build_problem_simplified(k0, N, loss, cons)
    qp = c.qp
    print(f"== {name} N={N} k0={k0} nz={qp.nz} nc={qp.nc} build {time.time()-t0:.2f}s plan {c._native.plan_info()}", flush=True)
    Bn = 8
    x0 = np.tile(zon.X0.center, (Bn, 1)) + 0.05 * rng.standard_normal((Bn, qp.n))
    e0 = 0.02 * rng.standard_normal((Bn, qp.n)); e0[0] = 0
    t0 = time.time()
    out = c.solve_batch(x0, e0, want_active=True)
    print(f"   solve_batch {time.time()-t0:.3f}s status {out['status']} iters {out['iters']}", flush=True)
    D, E, cs = c._scal
    for b in range(min(Bn, 3)):
        th_dev = c._native.debug_fetch(b, 0)
        th = theta_reference(qp, x0[b], e0[b])
        q_dev = c._native.debug_fetch(b, 1); h_dev = c._native.debug_fetch(b, 2)
        ql = qp.q0 + qp.Qt @ th; ll = qp.l0 + qp.Lt @ th; ul = qp.u0 + qp.Ut @ th
        print(f"   b={b} theta err {np.abs(th_dev-th).max():.1e} q err {np.abs(q_dev - cs*D*ql).max():.1e}", end=" ")
        r = solve_qp(qp.P, ql, qp.A, ll, ul, tol=1e-12)
        const = qp.r0 + qp.r1 @ x0[b] + x0[b] @ qp.R2 @ x0[b]
        nv = N * qp.m
        xb = (qp.Phi @ x0[b] + qp.Gam @ r.x[:nv]).reshape(N + 1, qp.n)
        print(f"| oracle {r.status} obj {r.obj+const:.10g} dev {out['cost'][b]:.10g} | dv0 {np.abs(out['v'][b,0]-r.x[:qp.m]).max():.1e} dxbar1 {np.abs(out['xbar'][b,1]-xb[1]).max():.1e} dv[:-1] {np.abs(out['v'][b].ravel()[:nv-qp.m]-r.x[:nv-qp.m]).max():.1e}", flush=True)
