"""Systems, data generation, gain and losses restated from the reference examples -- ORACLE.

Test infrastructure only (see oracle/__init__.py).

  generate_trajectories   reference ``examples/utils.py:6-45`` (incl. the quirk that the returned
                          state row 0 of every trajectory is all-zero, ``:33,40,42``)
  systems                 ``examples/1.double_integrator_sim.py:37-52``,
                          ``examples/1.double_integrator_computation_complexity.py:38-56``,
                          ``examples/2.pulley_sim.py:39-54``, ``examples/3.5dimsystem_sim.py:29-49``
  identify                ``tzddpc/tzddpc.py:60-62, 81-83, 119-128``
  gain                    stand-in for ``tzddpc/utils.py:60-103`` (LMI+DCCP+MOSEK, out of scope): LQR on
                          the identified (A, B); K is an input fixture for parity.
  losses                  ``examples/1...sim.py:22-28``, ``examples/2.pulley_sim.py:17-22``,
                          ``examples/3.5dimsystem_sim.py:14-26``
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import scipy.signal as scipysig

from .literal import AffineLoss
from .zonolite import MatrixZonotope, Zonotope, compute_LTI_matrix_zonotope, concatenate_zonotope


def generate_trajectories(Asys, Bsys, X0: Zonotope, U: Zonotope, W: Zonotope, num_trajectories: int, num_steps: int, rng):
    n, m = Bsys.shape
    total = num_steps * num_trajectories
    u = U.sample(total, rng).reshape(num_trajectories, num_steps, m)
    Wv = W.compute_vertices()
    X = np.zeros((num_trajectories, num_steps, n)); Y = np.zeros_like(X)
    for j in range(num_trajectories):
        X[j, 0] = X0.sample(1, rng)[0]
        for i in range(1, num_steps):
            # reference :40 writes np.squeeze(B * u), which is B @ u for dim_u = 1 and a shape error for dim_u > 1 (no reference
            # example has more than one input): B @ u is what it computes where it runs
            X[j, i] = Asys @ X[j, i - 1] + Bsys @ u[j, i - 1] + Wv[rng.integers(len(Wv))]
            Y[j, i] = X[j, i]
    return u.reshape(total, m), Y.reshape(total, n)


def system(name: str):
    """-> dict(A, B, X0, U, W, X, T) for 'di_sim', 'di_cc' (complexity script), 'pulley', 'dim5'."""
    if name in ("di_sim", "di_cc"):
        A = np.array([[1.0, 1.0], [0.0, 1.0]]); B = np.array([[0.5], [1.0]])
        X0 = Zonotope([-5, -2], 0 * np.eye(2)); U = Zonotope([0], np.ones((1, 1)))
        if name == "di_sim":
            W = Zonotope(np.zeros(2), 0.1 * np.array([[1, 0.5], [0.5, 1]])); X = Zonotope([-4, 0], 0.95 * np.diag([5, 2.5]))
        else:
            W = Zonotope(np.zeros(2), 0.001 * np.array([[1, 0.5], [0.5, 1]])); X = Zonotope([-4, 0], 1.2 * np.diag([5, 2.5]))
        return dict(A=A, B=B, X0=X0, U=U, W=W, X=X, T=100)
    if name == "pulley":
        sys_ = scipysig.TransferFunction([0.28261, 0.50666], [1, -1.41833, 1.58939, -1.31608, 0.88642], dt=0.05).to_ss()
        A, B = np.asarray(sys_.A), np.asarray(sys_.B)
        n, m = B.shape
        return dict(A=A, B=B, X0=Zonotope([0] * n, np.zeros((n, 1))), U=Zonotope([1] * m, 3 * np.ones((m, 1))),
                    W=Zonotope([0] * n, 0.1 * np.ones((n, 1))), X=Zonotope([1] * n, 2 * np.ones((n, 1))), T=400)
    if name.startswith("dim5"):
        # 'dim5' / 'dim5_w001': the reference example (one input); 'dim5m2' / 'dim5m2_w001': BASELINE.json configs[3] as stated
        # (n = 5, m = 2) -- SURVEY.md 8d's synthetic second input column (1, 0, 1, 0, 1)', everything else as the example
        # (U = <[7] * dim_u, 100 I> is already written for any dim_u, examples/3.5dimsystem_sim.py:44)
        Ac = np.array([[-1, -4, 0, 0, 0], [4, -1, 0, 0, 0], [0, 0, -3, 1, 0], [0, 0, -1, -3, 0], [0, 0, 0, 0, -2.0]])
        Bc = np.ones((5, 1)) if "m2" not in name else np.array([[1.0, 1.0], [1.0, 0.0], [1.0, 1.0], [1.0, 0.0], [1.0, 1.0]])
        m = Bc.shape[1]
        A, B, _, _, _ = scipysig.cont2discrete((Ac, Bc, np.eye(5), 0 * Bc), dt=0.05)
        wscale = 0.01 if name.endswith("_w001") else 0.1
        Id = 20 * np.ones((5, 1)); Id[1] = 19
        return dict(A=A, B=B, X0=Zonotope([-2, 4, 3, -2.5, 5.5], np.zeros((5, 5))), U=Zonotope([7] * m, 100 * np.eye(m)),
                    W=Zonotope([0] * 5, wscale * np.ones((5, 1))), X=Zonotope([1, 20, 1, 1, 1], Id), T=400)
    if name == "di2in":
        # two-input double integrator (no counterpart in the reference: the second m >= 2 system of the parity suite): a force
        # input and a direct position actuator, complexity-script W / X, U = <0, diag(1, 0.5)>
        A = np.array([[1.0, 1.0], [0.0, 1.0]]); B = np.array([[0.5, 0.4], [1.0, 0.0]])
        return dict(A=A, B=B, X0=Zonotope([-5, -2], 0 * np.eye(2)), U=Zonotope([0, 0], np.diag([1.0, 0.5])),
                    W=Zonotope(np.zeros(2), 0.001 * np.array([[1, 0.5], [0.5, 1]])), X=Zonotope([-4, 0], 1.2 * np.diag([5, 2.5])), T=100)
    raise KeyError(name)


def lqr_gain(A, B, Q=None, R=None):
    n, m = B.shape
    Q = np.eye(n) if Q is None else Q; R = np.eye(m) if R is None else R
    Pm = sla.solve_discrete_are(A, B, Q, R)
    return -np.linalg.solve(R + B.T @ Pm @ B, B.T @ Pm @ A)


def identify(u, x, W: Zonotope, K=None):
    """reference ``tzddpc/tzddpc.py:60-62`` (split), ``:81-83`` (Mdata), ``:119-128`` (MdataK, Mdelta, reduce(1))."""
    Xm, Xp, Um = x[:-1], x[1:], u[:-1]
    n = x.shape[1]
    Mw = concatenate_zonotope(W, Xm.shape[0])
    Mdata = compute_LTI_matrix_zonotope(Xm, Xp, Um, Mw)
    Ahat, Bhat = Mdata.center[:, :n], Mdata.center[:, n:]
    if K is None:
        K = lqr_gain(Ahat, Bhat)
    MdataK = Mdata * np.vstack([np.eye(n), K])
    Mdelta = Mdata + (-1.0 * Mdata.center)
    return dict(Mdata=Mdata.reduce(1), MdataK=MdataK.reduce(1), Mdelta=Mdelta.reduce(1), K=K, A=Ahat, B=Bhat,
                MdataK_raw=MdataK, Mdelta_raw=Mdelta)


# ---- losses / constraints in index form: (nxi, x_idx (rows x n), u_idx (rows x m) or None) ----

def _sel(nxi, idx):
    F = np.zeros((len(idx), nxi)); F[np.arange(len(idx)), idx] = 1.0
    return F


def loss_di(nxi, x_idx, u_idx):
    """sum_i ||x_i||^2 + 1e-2 |u_i|_1 over i < horizon = u.shape[0]; u free -> its term is 0."""
    L = AffineLoss()
    H = x_idx.shape[0] - 1 if u_idx is None else u_idx.shape[0]
    for i in range(H):
        L.sq.append((1.0, _sel(nxi, x_idx[i]), np.zeros(x_idx.shape[1])))
        if u_idx is not None:
            for j in u_idx[i]:
                f = np.zeros(nxi); f[j] = 1.0
                L.ab.append((1e-2, f, 0.0))
    return L


def loss_pulley(nxi, x_idx, u_idx):
    L = AffineLoss()
    H = x_idx.shape[0] - 1 if u_idx is None else u_idx.shape[0]
    for i in range(H):
        f = np.zeros(nxi); f[x_idx[i, 0]] = 1.0
        L.ab.append((1.0, f, -1.0))
    return L


def loss_dim5(nxi, x_idx, u_idx):
    L = AffineLoss()
    H = x_idx.shape[0] - 1 if u_idx is None else u_idx.shape[0]
    for i in range(H):
        f = np.zeros(nxi); f[x_idx[i, 1]] = 1.0
        L.ab.append((1e9, f, -2.0))
        if u_idx is not None:           # ||u_i||_2 with m = 1 is |u_i|; with m > 1 a second-order cone (use loss_dim5_l1)
            assert u_idx.shape[1] == 1
            g = np.zeros(nxi); g[u_idx[i, 0]] = 1.0
            L.ab.append((1e-1, g, 0.0))
    return L


DIM5_TARGET = np.array([0.0, 3.0, 0.0, 0.0, 0.0])


def loss_dim5_quadratic(nxi, x_idx, u_idx):
    """sum_i ||x_i - DIM5_TARGET||^2 + 1e-2 |u_i|_1: a strictly convex loss on the 5-dim systems (not from the reference: with two
    inputs the example's loss leaves v non-unique -- it prices x[i,1] only -- so trajectories can be compared only on a loss
    that determines them)."""
    L = AffineLoss()
    H = x_idx.shape[0] - 1 if u_idx is None else u_idx.shape[0]
    for i in range(H):
        L.sq.append((1.0, _sel(nxi, x_idx[i]), -DIM5_TARGET))
        if u_idx is not None:
            for j in u_idx[i]:
                f = np.zeros(nxi); f[j] = 1.0
                L.ab.append((1e-2, f, 0.0))
    return L


def loss_dim5_l1(nxi, x_idx, u_idx):
    """The 5-dim loss with ||u_i||_1 in place of ||u_i||_2 (examples/3.5dimsystem_sim.py:19): the stated substitution for
    dim_u > 1 where the loss sees v (build_problem_simplified); identical to loss_dim5 when dim_u = 1 or u is free."""
    L = AffineLoss()
    H = x_idx.shape[0] - 1 if u_idx is None else u_idx.shape[0]
    for i in range(H):
        f = np.zeros(nxi); f[x_idx[i, 1]] = 1.0
        L.ab.append((1e9, f, -2.0))
        if u_idx is not None:
            for j in u_idx[i]:
                g = np.zeros(nxi); g[j] = 1.0
                L.ab.append((1e-1, g, 0.0))
    return L


def constraints_dim5(nxi, x_idx, u_idx):
    rows = []
    for i in range(x_idx.shape[0]):
        a = np.zeros(nxi); a[x_idx[i, 1]] = 1.0
        rows.append((a, 2.0, 10.0))
    return rows
