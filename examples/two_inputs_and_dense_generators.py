"""Round-3 additions in one script (needs an MI355X):

1. the 5-dim example of the reference (examples/3.5dimsystem_sim.py) with TWO inputs -- BASELINE.json's config 4 as stated.  The
   callbacks are the reference's, unchanged: its `1e-1 * cp.norm(u[i], p=2)` is a second-order cone for dim_u = 2, but in
   `build_problem` it sits on the free variable `u` and vanishes;
2. matrix zonotopes with DENSE generators (Girard order 2 instead of the boxes of `reduce(1)`) at horizon 20: 684 epigraph variables
   in the literal form, solved in cutting-plane form (the literal tubes are evaluated on the device, generator by generator).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tzddpc_amd import TZDDPC, cplite as cp
from tzddpc_amd.dist import vertex_noise
from tzddpc_amd.harness import generate_trajectories, system


def loss_5dim(u, x):                      # reference examples/3.5dimsystem_sim.py:14-20
    cost = 0
    for i in range(u.shape[0]):
        cost += 1e9 * cp.norm(x[i, 1] - 2, p=2) + 1e-1 * cp.norm(u[i], p=2)
    return cost


def constraints_5dim(u, x):               # :23-26
    return [x[:, 1] <= 10, x[:, 1] >= 2]


def loss_di(u, x):                        # reference examples/1.double_integrator_sim.py:22-28
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(x[i, :], p=2) ** 2 + 1e-2 * cp.norm(u[i], p=1)
    return cost


if __name__ == "__main__":
    # ---- 1. two inputs -------------------------------------------------------------------------------------------------------
    A, B, zon, T = system("dim5m2_w001")
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(25)))
    ctl.build_zonotopes_theta(zon)
    ctl.build_problem(20, loss_5dim, constraints_5dim)
    Bn, steps = 256, 20
    sim = ctl.simulate_batch(np.tile(zon.X0.center, (Bn, 1)), vertex_noise(zon.W.compute_vertices(), 0, Bn, steps), A, B)
    print(f"5-dim system, {B.shape[1]} inputs, horizon 20: {Bn} closed loops x {steps} steps, all solved {bool((sim['status'] == 0).all())}; "
          f"x[1] after two steps in [{sim['x'][:, 2:, 1].min():.3f}, {sim['x'][:, 2:, 1].max():.3f}] (target 2)")

    # ---- 2. dense generators, cutting planes ----------------------------------------------------------------------------------
    A, B, zon, T = system("di_cc")
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(25)))
    ctl.build_zonotopes(zon)
    ctl.compute_theta()
    n = ctl.dim_x
    ctl.MdataK = (ctl.Mdata * np.vstack([np.eye(n), ctl.theta.K])).reduce(2)          # order 2: twice as many generators, dense
    ctl.Mdelta = (ctl.Mdata + (-1.0 * ctl.Mdata.center)).reduce(2)
    ctl.Mdata = ctl.Mdata.reduce(1)
    ctl.build_problem_simplified(1, 20, loss_di, lambda u, x: [])                     # dense="auto": cutting-plane form
    rng = np.random.default_rng(1)
    x0 = np.tile(zon.X0.center, (64, 1)) + 0.3 * rng.standard_normal((64, 2)); e0 = 0.02 * rng.standard_normal((64, 2))
    out = ctl.solve_batch(x0, e0)
    tb = ctl.literal_tubes(e0, out["xbar"], out["v"])
    Xi = zon.X.interval
    worst = max((out["xbar"][:, :20] + tb["center"] + tb["rad_x"] - Xi.right_limit).max(), (Xi.left_limit - (out["xbar"][:, :20] + tb["center"] - tb["rad_x"])).max())
    print(f"double integrator, order-2 generators, k0 = 1, horizon 20: {ctl.qp.nz} variables instead of 684, {ctl.num_cuts()} sign patterns after "
          f"{ctl.cut_rounds} round(s); all solved {bool((out['status'] == 0).all())}; largest violation of the LITERAL state tubes {worst:.1e}")
