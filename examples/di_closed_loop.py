"""Double-integrator closed loop: the TZDDPC half of reference examples/1.double_integrator_sim.py:20-95 on the MI355X path.

Only the import lines differ from the reference script (cvxpy -> cplite, pyzonotope/utils -> tzddpc_amd); the callbacks,
zonotopes, `build_problem(2, ...)` and the loop body are the reference's.  Plotting / the ZPC comparator are out of scope.
"""
import os
import sys

import numpy as np
import scipy.signal as scipysig

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tzddpc_amd import cplite as cp                      # reference: import cvxpy as cp
from tzddpc_amd import TZDDPC, SystemZonotopes, Zonotope  # reference: from tzddpc import ...; from pyzonotope import Zonotope
from tzddpc_amd.harness import generate_trajectories      # reference: from utils import generate_trajectories

np.random.seed(25)


def loss_callback(u, x):
    horizon, dim_u, dim_x = u.shape[0], u.shape[1], x.shape[1]
    cost = 0
    for i in range(horizon):
        cost += cp.norm(x[i, :], p=2) ** 2 + 1e-2 * cp.norm(u[i], p=1)
    return cost


def constraints_callback(u, x):
    return []


def main(total_steps=12, verbose=True):
    A = np.array([[1, 1], [0, 1]]); B = np.array([[0.5], [1]])
    dim_x, dim_u = B.shape
    sys_ = scipysig.StateSpace(A, B, np.eye(dim_x), np.zeros((dim_x, dim_u)), dt=1)     # reference :37-40
    X0 = Zonotope([-5, -2], 0 * np.eye(dim_x))
    U = Zonotope([0], 1 * np.ones((1, 1)))
    W = Zonotope(np.zeros(dim_x), 0.1 * np.array([[1, 0.5], [0.5, 1]]))
    X = Zonotope([-4, 0], 0.95 * np.diag([5, 2.5]))
    zonotopes = SystemZonotopes(X0, U, X, W)
    W_vertices = W.compute_vertices()
    data = generate_trajectories(sys_, X0, U, W, 1, 100)                                    # reference :59, same call
    x0 = X0.sample().flatten()

    tzddpc = TZDDPC(data)
    tzddpc.build_zonotopes_theta(zonotopes)
    x = [x0]; xbar = [x[-1].copy()]; e = [np.zeros_like(x[-1])]
    Ze = [Zonotope(np.zeros(dim_x), np.zeros((dim_x, 1))) + x[-1]]
    tzddpc.build_problem(2, loss_callback, constraints_callback)
    for t in range(total_steps):
        result, v, xbark, Zek = tzddpc.solve(xbar[-1], e[-1], verbose=False)
        if verbose:
            print(f"[{t}] x: {x[-1]} - xbar: {xbar[-1]} - v: {v[0]}")
        xbar.append(xbark[1])
        u = tzddpc.theta.K @ e[-1] + v[0]
        x_next = A @ x[-1] + np.squeeze(B @ u) + W_vertices[np.random.choice(len(W_vertices))]
        x.append(x_next.flatten())
        e.append(x[-1] - xbar[-1])
        Zek = Zek.Z.value
        Ze.append(Zonotope(Zek[:, 0], Zek[:, 1:]) + xbar[-1])
    return np.array(x), np.array(xbar), Ze, zonotopes


if __name__ == "__main__":
    xs, xbars, Ze, _ = main()
    print("final state", xs[-1])
