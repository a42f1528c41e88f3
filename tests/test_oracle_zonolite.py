"""Known answers for the literal zonotope algebra (oracle) and for the product's host-side set objects.

Hand-computed 2-D cases; the generator-count recurrences of SURVEY.md section 8a-3 (Gamma(Ze[k]) = 1, 24, 64 for the double
integrator, 75 for the pulley, 113 for the 5-dim system) pin the literal stacking order of reference tzddpc/tzddpc.py:172-207.
"""
import numpy as np
import pytest

from oracle import zonolite as zl
from tzddpc_amd import zonotope as pz


@pytest.mark.parametrize("mod", [zl, pz])
def test_zonotope_known_answers(mod):
    Z = mod.Zonotope([1.0, -2.0], [[1.0, 0.5], [0.0, 2.0]])
    iv = Z.interval
    np.testing.assert_allclose(iv.left_limit, [1 - 1.5, -2 - 2.0])
    np.testing.assert_allclose(iv.right_limit, [1 + 1.5, -2 + 2.0])
    Z2 = Z + mod.Zonotope([0.5, 0.5], [[1.0], [1.0]])          # Minkowski sum
    assert Z2.num_generators == 3
    np.testing.assert_allclose(Z2.center, [1.5, -1.5])
    np.testing.assert_allclose(Z2.interval.right_limit, [1.5 + 2.5, -1.5 + 3.0])
    K = np.array([[2.0, -1.0]])
    ZK = Z * K                                                  # left multiplication K @ Z (reference :192)
    assert ZK.dimension == 1
    np.testing.assert_allclose(ZK.center, [4.0])
    np.testing.assert_allclose(np.abs(ZK.generators).sum(), 2.0 + 1.0)
    Zs = Z + np.array([1.0, 1.0])
    np.testing.assert_allclose(Zs.center, [2.0, -1.0])
    assert len(Z.compute_vertices()) == 4


@pytest.mark.parametrize("mod", [zl, pz])
def test_matrix_zonotope_product_and_girard(mod):
    C = np.array([[1.0, 2.0], [0.0, 1.0]])
    G1 = np.array([[0.1, 0.0], [0.0, 0.0]]); G2 = np.array([[0.0, 0.0], [0.0, 0.2]])
    M = mod.MatrixZonotope(C, np.array([G1, G2]))
    Z = mod.Zonotope([1.0, 1.0], [[0.5], [0.0]])
    R = M * Z                                                   # [C Z, G1 Z, G2 Z]: (2+1)(1+1)-1 = 5 generators
    assert R.num_generators == 5
    np.testing.assert_allclose(R.center, C @ [1.0, 1.0])
    np.testing.assert_allclose(np.abs(R.generators).sum(axis=1), [0.5 + 0.1 + 0.05, 0.2])
    MK = M * np.array([[1.0], [2.0]])
    assert MK.shape == (2, 1)
    np.testing.assert_allclose(MK.center, C @ [[1.0], [2.0]])
    # Girard order 1: more generators than dimensions -> one axis-aligned box
    Zg = mod.Zonotope([0.0, 0.0], [[1.0, 0.5, -0.25], [0.5, 1.0, 0.25]]).reduce(1)
    np.testing.assert_allclose(np.abs(Zg.generators).sum(axis=1), [1.75, 1.75])
    assert np.count_nonzero(Zg.generators) == 2


@pytest.mark.parametrize("mod", [zl, pz])
def test_identification_contains_true_system(mod):
    rng = np.random.default_rng(0)
    A = np.array([[1.0, 1.0], [0.0, 1.0]]); B = np.array([[0.5], [1.0]])
    W = mod.Zonotope([0, 0], 0.01 * np.eye(2))
    T = 30
    x = np.zeros((T, 2)); u = rng.uniform(-1, 1, (T, 1))
    for t in range(1, T):
        x[t] = A @ x[t - 1] + B @ u[t - 1] + 0.01 * rng.uniform(-1, 1, 2)
    Mw = mod.concatenate_zonotope(W, T - 1)
    assert Mw.num_generators == 2 * (T - 1)
    Md = mod.compute_LTI_matrix_zonotope(x[:-1], x[1:], u[:-1], Mw)
    assert Md.shape == (2, 3)
    assert Md.contains(np.hstack([A, B]))
    box = Md.reduce(1)
    assert box.num_generators == 6 and all(np.count_nonzero(g) <= 1 for g in box.generators)


def test_generator_counts_follow_reference_stacking():
    from oracle import harness as H, literal as L
    expect = {"di_sim": [1, 24, 64], "pulley": [1, 75], "dim5": [1, 113]}
    for name, loss, cons in (("di_sim", H.loss_di, None), ("pulley", H.loss_pulley, None), ("dim5", H.loss_dim5, H.constraints_dim5)):
        s = H.system(name)
        rng = np.random.default_rng(25)
        u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], rng)
        idn = H.identify(u, x, s["W"])
        N = len(expect[name])
        lp = L.build_literal(idn["A"], idn["B"], idn["MdataK"], idn["Mdelta"], idn["K"], s["W"], s["X"], s["U"], N,
                             np.zeros(s["B"].shape[0]), s["X0"].center, loss, cons)
        assert [z.num_generators for z in lp.Ze] == expect[name]


def test_trajectory_generator_quirk_first_row_zero():
    """reference examples/utils.py:33,40,42 -- the returned state row 0 is all-zero, not X0."""
    from oracle import harness as H
    from tzddpc_amd.harness import generate_trajectories, system
    A, B, zon, T = system("di_sim")
    d = generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, 20, np.random.default_rng(1))
    assert np.all(d.x[0] == 0.0) and np.any(d.x[1] != 0.0)
    s = H.system("di_sim")
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, 20, np.random.default_rng(1))
    assert np.all(x[0] == 0.0)
