"""Feedback-gain helpers (offline, one-shot; NOT on the per-step path).

The reference synthesises K by alternating an LMI feasibility problem with a DCCP/MOSEK
adversarial search (reference ``tzddpc/utils.py:43-103``); both need cvxpy, dccp and a commercial
MOSEK licence and return a solver-dependent feasible point.  That loop is out of scope
(SURVEY.md section 2 row 7): K is an input of the hot path.  Provided here:

  * ``lqr_gain``          default gain when the caller supplies none (Riccati on the identified model)
  * ``spectral_radius``   reference ``utils.py:8-11``
  * ``is_gain_robust``    reference ``utils.py:105-129`` (same sample-size formula, numpy only)
  * ``compute_theta``     passthrough / LQR + robustness check with the reference's signature
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .objects import Theta
from .zonotope import MatrixZonotope


def spectral_radius(X: np.ndarray) -> float:
    X = np.asarray(X)
    assert X.ndim == 2 and X.shape[0] == X.shape[1], "X is not  a square matrix"
    return float(np.abs(np.linalg.eigvals(X)).max())


def lqr_gain(A: np.ndarray, B: np.ndarray, Q: Optional[np.ndarray] = None, R: Optional[np.ndarray] = None) -> np.ndarray:
    from scipy.linalg import solve_discrete_are
    n, m = B.shape
    Q = np.eye(n) if Q is None else Q
    R = np.eye(m) if R is None else R
    S = solve_discrete_are(A, B, Q, R)
    return -np.linalg.solve(R + B.T @ S @ B, B.T @ S @ A)


def is_gain_robust(Mdata: MatrixZonotope, K: np.ndarray, accuracy: float, confidence: float,
                   rng: Optional[np.random.Generator] = None) -> bool:
    assert K.shape[1] == Mdata.shape[0], "Wrong dimensionality for K"
    assert 0 < accuracy < 1, "Accuracy should be in (0,1)"
    assert 0 < confidence < 1, "confidence should be in (0,1)"
    n = K.shape[1]
    num = int(np.ceil(np.log(1 / confidence) / np.log(1 / (1 - accuracy))))
    AB = Mdata.sample(num, rng)
    Acl = AB[:, :, :n] + AB[:, :, n:] @ K
    return bool(np.abs(np.linalg.eigvals(Acl)).max() < 1.0)


def compute_theta(Mdata: MatrixZonotope, A0: np.ndarray, B0: np.ndarray, tolerance: float = 1e-5,
                  initial_points: int = 10, max_iterations: int = 20, accuracy: float = 1e-2,
                  confidence: float = 1e-5, K: Optional[np.ndarray] = None) -> Theta:
    """Same signature as reference ``utils.py:60``; ``K`` may be supplied (fixture), else LQR."""
    Kn = lqr_gain(A0, B0) if K is None else np.atleast_2d(np.asarray(K, dtype=float))
    assert is_gain_robust(Mdata, Kn, accuracy, confidence), \
        f"K is not robust with accuracy-confidence of {accuracy, 1 - confidence}"
    return Theta(Kn, np.zeros_like(A0), np.zeros_like(B0))
