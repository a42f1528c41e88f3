"""CPU: the oracle's restatement of the gain synthesis (reference tzddpc/utils.py) and the host side of the product's."""
import itertools

import numpy as np
import pytest

from oracle import gain as OG
from oracle import harness as H
from oracle.zonolite import MatrixZonotope as OMZ


@pytest.mark.parametrize("sysname", ["di_cc", "pulley", "dim5_w001"])
def test_lmi_point_satisfies_the_reference_constraints(sysname):
    """reference utils.py:47-52: X >> 0 and [[X, AX+BZ], [(AX+BZ)', X]] >> 0, K = Z inv(X) -- oracle and product points."""
    from tzddpc_amd import gain as PG
    s = H.system(sysname)
    A, B = s["A"], s["B"]
    for K, X, Z in (OG.compute_control_gain(A, B), PG.lmi_point(A, B)):
        assert OG.lmi_margin(A, B, X, Z) > 0
        np.testing.assert_allclose(Z @ np.linalg.inv(X), K, atol=1e-10)
        assert OG.spectral_radius(A + B @ K) < 1
    np.testing.assert_allclose(PG.compute_control_gain(A, B), OG.compute_control_gain(A, B)[0], atol=1e-12)


def test_ccp_ascent_is_monotone_and_reaches_the_best_vertex_on_small_boxes():
    """compute_A_B (reference utils.py:13-41): the objective is convex, so the maximum over the box is at a vertex; every CCP
    step must not decrease it, fixed points are vertices, and with enough starts the best fixed point is the global vertex maximum
    (brute force over 2^(2 gamma) vertices)."""
    rng = np.random.default_rng(0)
    n, m, g = 3, 1, 4
    Md = OMZ(rng.standard_normal((n, n + m)), 0.3 * rng.standard_normal((g, n, n + m)))
    K = rng.standard_normal((m, n))
    M0, Hh = OG.adversary_generators(Md, K)
    assert Hh.shape == (2 * g, n, n)
    Hf = Hh.reshape(2 * g, -1)
    best = max(np.linalg.norm(M0.reshape(-1) + np.array(v) @ Hf) for v in itertools.product((-1.0, 1.0), repeat=2 * g))
    tops = []
    for b0 in rng.uniform(-1, 1, size=(64, 2 * g)):
        f_prev = np.linalg.norm(M0.reshape(-1) + b0 @ Hf)
        b = b0
        for _ in range(50):
            nb, f, _ = OG.ccp_ascent(M0, Hh, b, max_iter=1)
            assert f >= f_prev - 1e-12
            if np.array_equal(nb, b):
                break
            b, f_prev = nb, f
        assert set(np.abs(b)) == {1.0}
        tops.append(f)
    assert max(tops) <= best + 1e-12 and abs(max(tops) - best) <= 1e-12
    An, Bn, f = OG.compute_A_B(Md, K, rng.uniform(-1, 1, size=(64, 2 * g)))
    assert abs(np.linalg.norm(An + Bn @ K) - f) <= 1e-12


def test_robustness_sampling_rule_and_host_evaluation():
    """reference utils.py:119: N = ceil(log(1/conf) / log(1/(1-acc))) (1146 at the defaults of compute_theta); the product's host
    evaluation (device=None) agrees with the oracle on the same coefficients, for a robust and for a non-robust gain."""
    from tzddpc_amd import gain as PG
    from tzddpc_amd.zonotope import MatrixZonotope as PMZ
    assert OG.num_robust_samples(1e-2, 1e-5) == 1146 == PG.num_robust_samples(1e-2, 1e-5)
    rng = np.random.default_rng(1)
    s = H.system("di_cc")
    n = 2
    C = np.hstack([s["A"], s["B"]])
    G = 0.01 * rng.standard_normal((7, 2, 3))
    K = OG.compute_control_gain(s["A"], s["B"])[0]
    beta = rng.uniform(-1, 1, size=(1146, 7))
    for gain in (K, np.zeros_like(K)):
        o = OG.is_gain_robust(OMZ(C, G), gain, 1e-2, 1e-5, beta)
        assert PG.is_gain_robust(PMZ(C, G), gain, 1e-2, 1e-5, beta=beta) == o
    assert OG.is_gain_robust(OMZ(C, G), K, 1e-2, 1e-5, beta) and not OG.is_gain_robust(OMZ(C, G), 0 * K, 1e-2, 1e-5, beta)


def test_oracle_alternation_terminates_with_a_stabilising_gain():
    from oracle.zonolite import compute_LTI_matrix_zonotope, concatenate_zonotope
    s = H.system("pulley")
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], np.random.default_rng(25))
    Md = compute_LTI_matrix_zonotope(x[:-1], x[1:], u[:-1], concatenate_zonotope(s["W"], x.shape[0] - 1))
    n = 4
    K, dA, dB, log = OG.compute_theta(Md, Md.center[:, :n], Md.center[:, n:], np.random.default_rng(1))
    assert log[-1][0] < 1 and len(log) <= 20
    assert dA.shape == (4, 4) and dB.shape == (4, 1)
    # the adversarial pair lies in the interval hull of Mdata (independent coefficients per block, reference :26-32)
    rad = np.abs(np.asarray(Md.generators)).sum(axis=0)
    assert (np.abs(np.hstack([dA, dB])) <= rad + 1e-12).all()
