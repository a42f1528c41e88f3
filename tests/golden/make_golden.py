"""Generates tests/golden/*.npz with the ORACLE ONLY (provenance "oracle only": nothing under tzddpc_amd/ is imported).

    data        oracle.harness.generate_trajectories        (reference examples/utils.py:6-45, zero-first-row quirk)
    model       oracle.harness.identify                     (reference tzddpc/tzddpc.py:60-62, 81-83, 119-128)
    problem     oracle.collapsed.build_collapsed            (reference :172-207 / :283-324; == oracle.literal for N <= 4,
                                                             tests/test_oracle_collapse.py)
    solution    oracle.qp_ipm.solve_qp + KKT certificate    (reference :367)

The reference cannot be imported here (ModuleNotFoundError: cvxpy / pyzonotope / pydatadrivenreachability, see
oracle/__init__.py), so these are the build's own goldens.  Stored per case: the data set and
gain (inputs of the product), four (xbar0, e0) points, and for each the optimal v, xbar, cost, the tubes (c_k, rho^x_k, rho^u_k)
at the optimum, their e0-only parts (what the device's tube pass computes), and the active set / slacks / multipliers of the
tube rows indexed [k, component, upper|lower].  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import collapsed as C, harness as H   # noqa: E402
from oracle.qp_ipm import solve_qp                # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# name: (system, loss, constraints, horizon, k0)       -- the same table as tests/common.CASES, in oracle terms
CASES = {
    "di_n2": ("di_sim", H.loss_di, None, 2, None),
    "di_sim_n5": ("di_sim", H.loss_di, None, 5, None),           # BASELINE config 1 as stated (N = 5 on the simulation zonotopes)
    "di_n5": ("di_cc", H.loss_di, None, 5, None),
    "di_n20": ("di_cc", H.loss_di, None, 20, None),
    "di_n20_k1": ("di_cc", H.loss_di, None, 20, 1),
    "di_n20_k2": ("di_cc", H.loss_di, None, 20, 2),
    "pulley_n10": ("pulley", H.loss_pulley, None, 10, None),
    "dim5_n20": ("dim5_w001", H.loss_dim5, H.constraints_dim5, 20, None),
    # m = 2 (BASELINE.json configs[3] as stated: n = 5, m = 2; the reference example itself has one input) and a second two-input system
    "dim5m2_n20": ("dim5m2_w001", H.loss_dim5, H.constraints_dim5, 20, None),
    "dim5m2q_n20": ("dim5m2_w001", H.loss_dim5_quadratic, H.constraints_dim5, 20, None),     # strictly convex loss: v, xbar comparable
    "di2in_n10": ("di2in", H.loss_di, None, 10, None),
    "di2in_n10_k1": ("di2in", H.loss_di, None, 10, 1),
}


def identified(sysname, seed=25):
    s = H.system(sysname)
    rng = np.random.default_rng(seed)
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], rng)
    return s, u, x, H.identify(u, x, s["W"])


# N = 5 with the simulation example's W = 0.1 is only just feasible from the corner X0: sampled a little inside
SAMPLING = {"di_sim_n5": dict(e_scale=0.005, x_scale=0.05, x_shift=0.4)}


def sample_params(s, B, seed=11, e_scale=0.02, x_scale=0.05, x_shift=0.0):
    n = s["B"].shape[0]
    rng = np.random.default_rng(seed)
    x0 = np.tile(s["X0"].center, (B, 1)) + x_shift + x_scale * rng.standard_normal((B, n))
    e0 = e_scale * rng.standard_normal((B, n))
    e0[0] = 0.0
    return x0, e0


def solve_point(s, idn, N, k0, loss, cons, x0, e0, tol=1e-12):
    cq = C.build_collapsed(idn["A"], idn["B"], idn["MdataK"], idn["Mdelta"], idn["K"], s["W"], s["X"], s["U"], N, e0, x0, loss, cons, k0)
    r = solve_qp(cq["P"], cq["q"], cq["A"], cq["l"], cq["u"], tol=tol)
    assert r.status == "solved", r.status
    c = r.cert
    assert max(c["primal"], c["dual"], c["comp"]) < 1e-9, c
    v, xb = C.extract(cq, r.x)
    n, m = cq["n"], cq["m"]
    tubes = C.collapsed_radii(cq, r.x[:cq["nxi"]])
    Ax = cq["A"] @ r.x
    act = np.zeros((N, n + m, 2), bool); slack = np.zeros((N, n + m, 2)); y = np.zeros((N, n + m, 2))
    row = cq["tube_row0"]
    for k in range(N):
        for c_ in range(n + m):
            for side in range(2):                     # 0: upper row, 1: lower row
                sl = (cq["u"][row] - Ax[row]) if side == 0 else (Ax[row] - cq["l"][row])
                slack[k, c_, side] = sl; y[k, c_, side] = abs(r.y[row]); act[k, c_, side] = sl < abs(r.y[row])
                row += 1
    return dict(v=v, xbar=xb, cost=r.obj + cq["r"], cert=[c["primal"], c["dual"], c["comp"]],
                tube_c=np.array([t[0] for t in tubes]), tube_rx=np.array([t[1] for t in tubes]), tube_ru=np.array([t[2] for t in tubes]),
                e0_c=np.array([t[0] for t in cq["e0tube"]]), e0_rx=np.array([t[1] for t in cq["e0tube"]]), e0_ru=np.array([t[2] for t in cq["e0tube"]]),
                active=act, slack=slack, y=y)


def closed_loop(s, idn, N, k0, loss, cons, x0, noise):
    """examples/1.double_integrator_sim.py:75-90 with the oracle solving every step from scratch."""
    K = idn["K"]; A, B = s["A"], s["B"]
    Bn, T, n = noise.shape
    xs = np.zeros((Bn, T + 1, n)); us = np.zeros((Bn, T, B.shape[1])); cost = np.zeros((Bn, T))
    for b in range(Bn):
        x = x0[b].copy(); xbar = x.copy(); e = np.zeros(n)
        xs[b, 0] = x
        for t in range(T):
            sol = solve_point(s, idn, N, k0, loss, cons, xbar, e)
            u = K @ e + sol["v"][0]                                            # :84
            x = A @ x + B @ u + noise[b, t]                                    # :85
            xbar = sol["xbar"][1]; e = x - xbar                                # :83, :87
            xs[b, t + 1] = x; us[b, t] = u; cost[b, t] = sol["cost"]
    return xs, us, cost


def main(only=None):
    for case, (sysname, loss, cons, N, k0) in CASES.items():
        if only and case not in only:
            continue
        s, u, x, idn = identified(sysname)
        x0, e0 = sample_params(s, 4, **SAMPLING.get(case, {}))
        rec = dict(data_u=u, data_x=x, K=idn["K"], x0=x0, e0=e0)
        sols = [solve_point(s, idn, N, k0, loss, cons, x0[b], e0[b]) for b in range(4)]
        for key in sols[0]:
            rec[key] = np.array([sl[key] for sl in sols])
        rec["provenance"] = np.array("oracle only: oracle.harness.identify + oracle.collapsed.build_collapsed + oracle.qp_ipm.solve_qp")
        np.savez_compressed(os.path.join(OUT, f"{case}.npz"), **rec)
        print(case, "cost", rec["cost"], "max cert", np.max(rec["cert"]))
    if only:
        return
    # G5: closed-loop double integrator (config-1 analogue: sim zonotopes, N = 2, 12 steps), fixed vertex noise
    s, u, x, idn = identified("di_sim")
    Wv = s["W"].compute_vertices()
    noise = np.stack([Wv[np.random.Generator(np.random.PCG64(500 + i)).integers(len(Wv), size=12)] for i in range(3)])
    x0 = np.tile(s["X0"].center, (3, 1))
    xs, us, cost = closed_loop(s, idn, 2, None, H.loss_di, None, x0, noise)
    np.savez_compressed(os.path.join(OUT, "di_n2_closed_loop.npz"), data_u=u, data_x=x, K=idn["K"], x0=x0, noise=noise, x=xs, u=us, cost=cost,
                        provenance=np.array("oracle only"))
    print("closed loop final states", xs[:, -1])
    # longer oracle-only closed loops, one of them with two inputs: (system, loss, constraints, N, k0, trajectories, steps)
    for name, (sysname, loss, cons, N, k0, Bn, T) in {"di2in_n5_closed_loop": ("di2in", H.loss_di, None, 5, None, 2, 30),
                                                       "pulley_n4_closed_loop": ("pulley", H.loss_pulley, None, 4, None, 2, 40)}.items():
        s, u, x, idn = identified(sysname)
        Wv = s["W"].compute_vertices()
        noise = np.stack([Wv[np.random.Generator(np.random.PCG64(700 + i)).integers(len(Wv), size=T)] for i in range(Bn)])
        x0 = np.tile(s["X0"].center, (Bn, 1))
        xs, us, cost = closed_loop(s, idn, N, k0, loss, cons, x0, noise)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), data_u=u, data_x=x, K=idn["K"], x0=x0, noise=noise, x=xs, u=us, cost=cost,
                            horizon=np.array(N), provenance=np.array("oracle only"))
        print(name, "final states", xs[:, -1])


if __name__ == "__main__":
    main(sys.argv[1:])
