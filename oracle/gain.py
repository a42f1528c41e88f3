"""TEST INFRASTRUCTURE -- CPU restatement of the reference's gain synthesis (``tzddpc/utils.py``), numpy / scipy only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product never does.

Parity unpinned: the reference delegates both optimisation problems to external solvers (cvxpy's default SDP solver for the LMI
``:43-58``, DCCP + MOSEK for the adversarial search ``:13-41``), neither of which is available here, and both problems have
non-unique solutions (``Minimize(1)`` feasibility; a non-convex maximisation started from random points).  What CAN be restated
exactly and is restated here:

  * ``spectral_radius``                 ``:8-11``
  * the feasible set of the LMI         ``:47-52``   -> ``lmi_margin`` certifies any (X, Z) / K against it
  * the objective and feasible set of the adversarial search ``:14-34`` (independent beta_A, beta_B per generator) and the
    convex-concave step DCCP applies to it (linearise the convex objective at the current point, maximise over the box)
  * the sampling test                   ``:105-129``  (sample-size formula, strict ``>= 1`` rejection)
  * the alternation of ``compute_theta`` ``:60-103``  (order of updates, stopping rule, returned deltas)
"""
from __future__ import annotations

import numpy as np


def spectral_radius(X):
    X = np.asarray(X, float)
    assert X.ndim == 2 and X.shape[0] == X.shape[1], "X is not  a square matrix"
    return float(np.abs(np.linalg.eigvals(X)).max())


def lmi_margin(A, B, X, Z):
    """Smallest eigenvalue over the two constraints of ``:52``:  X >> 0,  [[X, AX+BZ], [(AX+BZ)', X]] >> 0."""
    F = A @ X + B @ Z
    P = np.block([[X, F], [F.T, X]])
    return float(min(np.linalg.eigvalsh(0.5 * (X + X.T)).min(), np.linalg.eigvalsh(0.5 * (P + P.T)).min()))


def compute_control_gain(A, B):
    """A point of the LMI's feasible set (``:43-58`` returns whichever one its solver lands on): Riccati solution S of the
    discrete LQR with unit weights, X = inv(S), Z = K X.  Returns (K, X, Z)."""
    from scipy.linalg import solve_discrete_are
    n, m = B.shape
    S = solve_discrete_are(A, B, np.eye(n), np.eye(m))
    K = -np.linalg.solve(np.eye(m) + B.T @ S @ B, B.T @ S @ A)
    X = np.linalg.inv(S)
    return K, X, K @ X


def adversary_generators(Mdata, K):
    """M0 and the 2 gamma directions of ``:26-32``: A-columns and B-columns of every generator move independently."""
    n = K.shape[1]
    C, G = np.asarray(Mdata.center, float), np.asarray(Mdata.generators, float)
    M0 = C[:, :n] + C[:, n:] @ K
    H = np.concatenate([G[:, :, :n], G[:, :, n:] @ K], axis=0)
    return M0, H


def ccp_ascent(M0, H, beta0, max_iter=100):
    """One start: iterate beta <- argmax over the box of the objective linearised at beta (= sign <M(beta), H_i>, ties keep
    beta_i) until nothing changes.  Returns (beta, ||M(beta)||_F, steps)."""
    beta = np.array(beta0, float)
    Hf = H.reshape(H.shape[0], -1)
    steps = 0
    for _ in range(max_iter):
        M = M0.reshape(-1) + beta @ Hf
        d = Hf @ M
        new = np.where(d > 0, 1.0, np.where(d < 0, -1.0, beta))
        steps += 1
        if np.array_equal(new, beta):
            break
        beta = new
    M = M0.reshape(-1) + beta @ Hf
    return beta, float(np.sqrt(M @ M)), steps


def compute_A_B(Mdata, K, beta0):
    """``:13-41`` with the CCP restated: best fixed point over the starting points beta0 (S x 2 gamma).  Returns (An, Bn, fro)."""
    n = K.shape[1]
    M0, H = adversary_generators(Mdata, K)
    best = None
    for b0 in np.atleast_2d(beta0):
        b, f, _ = ccp_ascent(M0, H, b0)
        if best is None or f > best[1]:
            best = (b, f)
    b, f = best
    g = Mdata.generators.shape[0]
    C, G = np.asarray(Mdata.center, float), np.asarray(Mdata.generators, float)
    An = C[:, :n] + np.tensordot(b[:g], G[:, :, :n], axes=(0, 0))
    Bn = C[:, n:] + np.tensordot(b[g:], G[:, :, n:], axes=(0, 0))
    return An, Bn, f


def num_robust_samples(accuracy, confidence):
    return int(np.ceil(np.log(1 / confidence) / np.log(1 / (1 - accuracy))))        # :119


def sampled_radii(Mdata, K, beta):
    """Spectral radius of A + B K for the samples C + sum_i beta[s, i] G_i (``:121-125``)."""
    n = K.shape[1]
    X = np.asarray(Mdata.center, float)[None] + np.tensordot(beta, np.asarray(Mdata.generators, float), axes=(1, 0))
    return np.abs(np.linalg.eigvals(X[:, :, :n] + X[:, :, n:] @ K)).max(axis=1)


def is_gain_robust(Mdata, K, accuracy, confidence, beta):
    assert K.shape[1] == Mdata.center.shape[0], "Wrong dimensionality for K"
    assert 0 < accuracy < 1 and 0 < confidence < 1
    assert beta.shape[0] == num_robust_samples(accuracy, confidence)
    return bool(not np.any(sampled_radii(Mdata, K, beta) >= 1.0))


def compute_theta(Mdata, A0, B0, rng, tolerance=1e-5, initial_points=10, max_iterations=20):
    """The alternation of ``:60-103`` (without the print-outs and the two asserts at the end): returns (K, deltaA, deltaB, log)."""
    n, m = B0.shape
    g = np.asarray(Mdata.generators).shape[0]
    An, Bn, Kn = A0.copy(), B0.copy(), np.zeros((m, n))
    prev, it, log = 0.0, 0, []
    while it < max_iterations:
        Kn = compute_control_gain(An, Bn)[0]
        An, Bn, f = compute_A_B(Mdata, Kn, rng.uniform(-1.0, 1.0, size=(initial_points, 2 * g)))
        lam = max(spectral_radius(An + Bn @ Kn), spectral_radius(A0 + B0 @ Kn))
        log.append((lam, f))
        if abs(lam - prev) < tolerance or lam < 1:
            break
        it += 1
        prev = lam
    return Kn, An - A0, Bn - B0, log
