"""Example-side helpers restated for the MI355X package (build-time / data generation only).

``generate_trajectories`` follows reference ``examples/utils.py:6-45`` including its quirk: the returned
state row 0 of every trajectory is all-zero (``Y`` is never assigned at i = 0, ``:33,40,42``), which
matters for parity of ``Mdata``.  ``system`` returns the plants / zonotopes of the reference examples.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import scipy.signal as scipysig

from .objects import Data, SystemZonotopes
from .zonotope import Zonotope


def generate_trajectories(*args, rng: Optional[np.random.Generator] = None) -> Data:
    """Two call forms:

    ``generate_trajectories(sys, X0, U, W, num_trajectories, num_steps)`` -- the reference's (``examples/utils.py:6-12``): ``sys``
    is anything with ``.A`` and ``.B`` (``scipy.signal.StateSpace``); random numbers from numpy's global state, as there;
    ``generate_trajectories(A, B, X0, U, W, num_trajectories, num_steps[, rng])`` -- plant matrices and an explicit generator.
    """
    if len(args) >= 6 and hasattr(args[0], "A") and hasattr(args[0], "B"):
        sysd, X0, U, W, num_trajectories, num_steps = args[:6]
        A, B = np.asarray(sysd.A, float), np.asarray(sysd.B, float)
        rest = args[6:]
    else:
        A, B, X0, U, W, num_trajectories, num_steps = args[:7]
        rest = args[7:]
    if rest:
        rng = rest[0]
    A = np.asarray(A, float); B = np.asarray(B, float)
    n, m = B.shape
    total = num_steps * num_trajectories
    u = U.sample(total, rng).reshape(num_trajectories, num_steps, m)
    Wv = W.compute_vertices()
    X = np.zeros((num_trajectories, num_steps, n)); Y = np.zeros_like(X)
    for j in range(num_trajectories):
        X[j, 0] = X0.sample(1, rng)[0]
        for i in range(1, num_steps):
            pick = np.random.choice(len(Wv)) if rng is None else rng.integers(len(Wv))
            # reference :40: np.squeeze(B * u) -- B @ u for dim_u = 1, a shape error for dim_u > 1 (no reference example has one)
            X[j, i] = A @ X[j, i - 1] + B @ u[j, i - 1] + Wv[pick]
            Y[j, i] = X[j, i]
    return Data(u.reshape(total, m), Y.reshape(total, n))


def system(name: str):
    """(A, B, SystemZonotopes, T) for 'di_sim' (examples/1.double_integrator_sim.py:37-52), 'di_cc'
    (examples/1.double_integrator_computation_complexity.py:38-56), 'pulley' (examples/2.pulley_sim.py:39-54),
    'dim5' / 'dim5_w001' (examples/3.5dimsystem_sim.py:29-49; W scaled to 0.01 for N = 20)."""
    if name in ("di_sim", "di_cc"):
        A = np.array([[1.0, 1.0], [0.0, 1.0]]); B = np.array([[0.5], [1.0]])
        X0 = Zonotope([-5, -2], 0 * np.eye(2)); U = Zonotope([0], np.ones((1, 1)))
        if name == "di_sim":
            W = Zonotope(np.zeros(2), 0.1 * np.array([[1, 0.5], [0.5, 1]])); X = Zonotope([-4, 0], 0.95 * np.diag([5, 2.5]))
        else:
            W = Zonotope(np.zeros(2), 0.001 * np.array([[1, 0.5], [0.5, 1]])); X = Zonotope([-4, 0], 1.2 * np.diag([5, 2.5]))
        return A, B, SystemZonotopes(X0, U, X, W), 100
    if name == "pulley":
        s = scipysig.TransferFunction([0.28261, 0.50666], [1, -1.41833, 1.58939, -1.31608, 0.88642], dt=0.05).to_ss()
        A, B = np.asarray(s.A), np.asarray(s.B)
        n, m = B.shape
        return A, B, SystemZonotopes(Zonotope([0] * n, np.zeros((n, 1))), Zonotope([1] * m, 3 * np.ones((m, 1))),
                                     Zonotope([1] * n, 2 * np.ones((n, 1))), Zonotope([0] * n, 0.1 * np.ones((n, 1)))), 400
    if name in ("dim5", "dim5_w001", "dim5m2", "dim5m2_w001"):
        # '...m2': BASELINE.json configs[3] as stated (n = 5, m = 2) -- SURVEY.md 8d's synthetic second input column (1,0,1,0,1)';
        # U = <[7] * dim_u, 100 I> is the example's own formula (examples/3.5dimsystem_sim.py:44)
        Ac = np.array([[-1, -4, 0, 0, 0], [4, -1, 0, 0, 0], [0, 0, -3, 1, 0], [0, 0, -1, -3, 0], [0, 0, 0, 0, -2.0]])
        Bc = np.ones((5, 1)) if "m2" not in name else np.array([[1.0, 1.0], [1.0, 0.0], [1.0, 1.0], [1.0, 0.0], [1.0, 1.0]])
        m = Bc.shape[1]
        A, B, _, _, _ = scipysig.cont2discrete((Ac, Bc, np.eye(5), 0 * Bc), dt=0.05)
        Id = 20 * np.ones((5, 1)); Id[1] = 19
        w = 0.01 if name.endswith("_w001") else 0.1
        return A, B, SystemZonotopes(Zonotope([-2, 4, 3, -2.5, 5.5], np.zeros((5, 5))), Zonotope([7] * m, 100 * np.eye(m)),
                                     Zonotope([1, 20, 1, 1, 1], Id), Zonotope([0] * 5, w * np.ones((5, 1)))), 400
    if name == "di2in":
        # two-input double integrator (no counterpart in the reference; second m >= 2 system of the parity suite)
        A = np.array([[1.0, 1.0], [0.0, 1.0]]); B = np.array([[0.5, 0.4], [1.0, 0.0]])
        return A, B, SystemZonotopes(Zonotope([-5, -2], 0 * np.eye(2)), Zonotope([0, 0], np.diag([1.0, 0.5])),
                                     Zonotope([-4, 0], 1.2 * np.diag([5, 2.5])),
                                     Zonotope(np.zeros(2), 0.001 * np.array([[1, 0.5], [0.5, 1]]))), 100
    raise KeyError(name)
