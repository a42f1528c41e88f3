"""Gain synthesis without MOSEK and the solve_simplified2 formulation (reference tzddpc/utils.py:60-103, tzddpc/tzddpc.py:381-500).

  python examples/gain_and_simplified2.py

1. build_zonotopes_theta(..., synthesize=True): the reference's alternation -- LMI point (Riccati), adversarial (A, B) by
   convex-concave sign updates from 10 starting points on the GPU, robustness test on 1146 sampled closed loops on the GPU.
2. solve_simplified2 with constant tube generators W + Zsigma and the adversarial model errors of step 1.
"""
import os
import sys

import numpy as np
import scipy.signal as scipysig

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tzddpc_amd import TZDDPC, SystemZonotopes, Zonotope, cplite as cp, spectral_radius
from tzddpc_amd.harness import generate_trajectories

np.random.seed(25)
A = np.array([[1.0, 1.0], [0.0, 1.0]]); B = np.array([[0.5], [1.0]])
sys_ = scipysig.StateSpace(A, B, np.eye(2), np.zeros((2, 1)), dt=1)
W = Zonotope(np.zeros(2), 0.01 * np.array([[1, 0.5], [0.5, 1]]))
X = Zonotope([-4, 0], 0.95 * np.diag([5, 2.5])); U = Zonotope([0], np.ones((1, 1))); X0 = Zonotope([-5, -2], np.zeros((2, 2)))
zonotopes = SystemZonotopes(X0, U, X, W)

ctl = TZDDPC(generate_trajectories(sys_, X0, U, W, 1, 100))
theta, Mdata = ctl.build_zonotopes_theta(zonotopes, synthesize=True, rng=np.random.default_rng(1))
Ahat, Bhat = Mdata.center[:, :2], Mdata.center[:, 2:]
print("K =", theta.K.ravel(), " spectral radius of the identified loop:", round(spectral_radius(Ahat + Bhat @ theta.K), 4))
print("adversarial model errors  dA =", np.round(theta.deltaA.ravel(), 4), " dB =", np.round(theta.deltaB.ravel(), 4))


def loss(u, x):
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(x[i, :], p=2) ** 2 + 1e-2 * cp.norm(u[i], p=1)
    return cost


N = 6
Zsigma = [Zonotope(np.zeros(2), 0.005 * np.eye(2)) for _ in range(N)]
x = np.array([-3.0, -0.5]); xbar = x.copy(); e = np.zeros(2)
for t in range(8):
    result, v, xb, Ze1 = ctl.solve_simplified2(xbar, e, N, Zsigma, loss, lambda u, x: [])
    u = theta.K @ x + v[0]                     # ubar_0 = K xbar_0 + v_0 is the nominal input (:428); the error feedback acts on x - xbar through K
    x = A @ x + B @ u + W.sample()[0]
    xbar = xb[1]; e = x - xbar
    print(f"[{t}] cost {result:9.4f}  x {np.round(x, 4)}  xbar {np.round(xbar, 4)}  |Ze1 generators| {Ze1.Z.value.shape[1] - 1}")
