"""CPU tests on the reference's stored double-integrator figure (tests/refdi.py explains what its numbers determine): the artefact's
internal consistency, the tube formula against its polygons, and the oracle chain and the product's builder against the inputs the
reference applied -- in particular in the four steps where tightened tube rows are ACTIVE."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import common, refdi  # noqa: E402
from tests.test_oracle_golden import _make_golden  # noqa: E402


@pytest.fixture(scope="module")
def fig():
    g = refdi.vectors()
    m = refdi.recover_model_and_inputs(g)
    f = refdi.fit_tube_constants(g, m)
    return g, m, f


def _inside(poly, p, tol=1e-9):
    """p inside the convex polygon (vertices in order)?"""
    d = np.roll(poly, -1, axis=0) - poly
    c = d[:, 0] * (p[1] - poly[:, 1]) - d[:, 1] * (p[0] - poly[:, 0])
    return bool(np.all(c >= -tol) or np.all(c <= tol))


def test_figure_vectors_are_what_the_artefact_says(fig):
    g, m, f = fig
    assert str(g["provenance"]).startswith("reference artefact")
    x, P = g["x"], g["polygons"]
    np.testing.assert_allclose(x[0], [-5.0, -2.0], atol=1e-8)                        # examples/1.double_integrator_sim.py:49
    # one candidate sequence explains the polygon centres, no other comes close; the model it identifies is near the true plant
    assert m["ambiguous_steps"] == 10
    assert m["residual"] <= 5e-8 and m["runner_up"] >= 0.05, (m["residual"], m["runner_up"])
    assert np.abs(m["Ahat"] - refdi.A_TRUE).max() <= 0.01 and abs(m["Bhat"][1] - 1.0) <= 0.05 and abs(m["Bhat"][0] - 0.5) <= 0.2
    assert np.abs(m["u"]).max() <= 1.0 + 1e-7 and abs(m["u"][2] - 1.0) <= 1e-7       # U = [-1, 1] (:50); saturated at step 2
    # every polygon is W + a box: edges 2 g_1 and 2 g_2 of W to the last digit, one horizontal, one vertical, same half-width twice
    for t in range(1, 13):
        ed = np.roll(P[t], -1, axis=0) - P[t]
        ed = ed[np.argsort(np.arctan2(ed[:, 1], ed[:, 0]) % np.pi)][::2]                 # the four edge directions, by angle in [0, pi)
        assert abs(ed[0][1]) <= 1e-9 and abs(ed[3][0]) <= 1e-9                           # horizontal, ..., vertical
        np.testing.assert_allclose(np.abs(ed[1]), 2 * refdi.W_GEN[:, 0], atol=2e-7)
        np.testing.assert_allclose(np.abs(ed[2]), 2 * refdi.W_GEN[:, 1], atol=2e-7)
        assert abs(abs(ed[0][0]) - abs(ed[3][1])) <= 2e-7
        assert _inside(P[t], x[t])                                                    # the tube of step t-1 holds the state that followed
    assert abs(P[4][:, 1].max() - 2.0) <= 5e-7                                       # the stage-1 state tube at its bound (step 3)
    assert P[:, :, 0].min() > refdi.X_LOW[0] and P[:, :, 0].max() < refdi.X_HIGH[0] and P[:, :, 1].min() > refdi.X_LOW[1]


def test_tube_formula_explains_the_polygon_sizes(fig):
    g, m, f = fig
    assert f["residual"] <= 1e-8, f["residual"]                                      # 12 equations, 7 unknowns
    assert np.all(f["dK"] > 0) and np.all(f["dD"] > 0)
    assert np.all(f["sigma"] <= 1e-6)                                                # what the figure's rounding leaves of the parameters
    Ah, Bh, CK, DK, DD, K = refdi.model_matrices(m, f)
    assert np.abs(np.linalg.eigvals(CK)).max() < 1.0                                 # the reference's gain stabilises its model
    # the centre of every polygon is the model's one-step prediction from the TRUE state and the applied input (the gain cancels)
    c, half = refdi.polygon_centres_and_halfwidths(g)
    np.testing.assert_allclose(c[1:], g["x"][:-1] @ Ah.T + np.outer(m["u"], Bh[:, 0]), atol=5e-8)
    # ... and equals xbar_{t+1} + CK e_t with the recovered nominal chain
    np.testing.assert_allclose(c[1:], f["xbar"][1:] + f["e"] @ CK.T, atol=5e-8)
    # the recovered box behaves like the Mdata of tzddpc/tzddpc.py:81-83, :119-128: the true plant lies inside it (the data-driven
    # guarantee), the box of MdataK obeys the interval bound of Mdata [I; K], and the reference's gain (its LMI / CCP synthesis,
    # tzddpc/utils.py:13-129) stabilises every vertex of the box -- by the reference's own test (`is_gain_robust`, :105-129), on samples
    import itertools
    from tzddpc_amd.gain import is_gain_robust
    from tzddpc_amd.zonotope import boxed_matrix_zonotope
    M0 = np.hstack([Ah, Bh])
    assert np.all(np.abs(np.hstack([refdi.A_TRUE, refdi.B_TRUE[:, None]]) - M0) <= DD)
    assert np.all(DK[0] <= DD[0, :2] + DD[0, 2] * np.abs(K[0]) + 1e-9)
    worst = max(np.abs(np.linalg.eigvals(M[:, :2] + M[:, 2:] @ K)).max()
                for M in (M0 + np.array(sg, float).reshape(2, 3) * DD for sg in itertools.product((1, -1), repeat=6)))
    assert worst < 0.7, worst
    assert is_gain_robust(boxed_matrix_zonotope(M0, DD), K, 0.05, 0.99, rng=np.random.default_rng(3))


def _oracle_model(m, f):
    """idn / s dictionaries of the oracle chain (tests/golden/make_golden.solve_point) for the figure's run."""
    from oracle import harness as H
    from oracle.zonolite import MatrixZonotope, Zonotope
    Ah, Bh, CK, DK, DD, K = refdi.model_matrices(m, f)

    def boxed(center, mags):
        gens = []
        for r in range(mags.shape[0]):
            for c_ in range(mags.shape[1]):
                G = np.zeros_like(center); G[r, c_] = mags[r, c_]; gens.append(G)
        return MatrixZonotope(center, np.array(gens))

    s = H.system("di_sim")
    s = dict(s); s["X"] = Zonotope(0.5 * (refdi.X_LOW + refdi.X_HIGH), np.diag(0.5 * (refdi.X_HIGH - refdi.X_LOW)))
    idn = dict(A=Ah, B=Bh, K=K, MdataK=boxed(CK, DK), Mdelta=boxed(np.zeros((2, 3)), DD))
    return H, s, idn


def _oracle_loss_on_v(nxi, x_idx, u_idx):
    """|xbar_1|^2 + 1e-2 (|v_0| + |v_1|) in the oracle's index form (simplified-problem convention: x_idx = rows of xbar[1:])."""
    from oracle.harness import AffineLoss, _sel
    L = AffineLoss()
    L.sq.append((1.0, _sel(nxi, x_idx[0]), np.zeros(x_idx.shape[1])))
    for i in range(u_idx.shape[0]):
        for j in u_idx[i]:
            fvec = np.zeros(nxi); fvec[j] = 1.0
            L.ab.append((1e-2, fvec, 0.0))
    return L


def _check_steps(solve, m, f, penalised):
    """`solve(xbar, e) -> (v0, xbar1)` at the reference's own twelve operating points; returns the per-step u - u_ref."""
    K = f["K"]
    d = np.zeros(12)
    for t in range(12):
        v0, xb1 = solve(f["xbar"][t], f["e"][t])
        d[t] = float(np.ravel(K @ f["e"][t] + v0)[0]) - m["u"][t]
        if t < 4 or penalised:
            np.testing.assert_allclose(xb1, f["xbar"][t + 1], atol=refdi.TOL_ACTIVE if t < 4 else refdi.TOL_LOOP)
    assert np.abs(d[:4]).max() <= refdi.TOL_ACTIVE, d[:4]                             # (a): tightened tube rows active
    if penalised:
        assert np.abs(d).max() <= refdi.TOL_LOOP, d                                   # (c)
    else:
        np.testing.assert_allclose(d[[4, 6, 7, 8, 9, 10, 11]], -refdi.l1_offset(m), atol=refdi.TOL_LOOP)   # (b); step 5: see below
    return d


def test_oracle_chain_reproduces_the_reference_inputs(fig):
    """oracle.collapsed + oracle.qp_ipm (KKT certificate) on the figure's run: committed formulation (free u) and penalty on v."""
    g, m, f = fig
    mg = _make_golden()
    H, s, idn = _oracle_model(m, f)
    sols = {}

    def committed(xb, e):
        sol = mg.solve_point(s, idn, 2, None, H.loss_di, None, xb, e); sols[len(sols)] = sol
        return sol["v"][0], sol["xbar"][1]
    d0 = _check_steps(committed, m, f, penalised=False)
    # the tube rows that are active in steps 0 .. 3: stage-1 input tube on BOTH sides (it fills U), input bound, stage-1 state tube
    act = [sols[t]["active"] for t in range(4)]                                       # (N, n + m, side)
    assert act[0][1, 2].all() and act[1][1, 2].all() and act[2][0, 2, 0] and act[3][1, 1, 0]
    assert not any(sols[t]["active"].any() for t in (4, 6, 7, 8, 9, 10, 11))
    d1 = _check_steps(lambda xb, e: (lambda sol: (sol["v"][0], sol["xbar"][1]))(mg.solve_point(s, idn, 2, 2, _oracle_loss_on_v, None, xb, e)),
                      m, f, penalised=True)
    # step 5 with the penalty on v has the stage-1 input tube active on one side, which the committed formulation (no pull on v_1) has
    # not: its offset there is not the unconstrained one
    assert abs(d0[5] + refdi.l1_offset(m)) > 1e-3 and abs(d1[5]) <= refdi.TOL_LOOP


def _product_qp(m, f, penalised):
    from tzddpc_amd import cplite as cp
    from tzddpc_amd.builder import build_parametric_qp
    from tzddpc_amd.harness import system
    _, _, zon, _ = system("di_sim")
    Ah, Bh, CK, DK, DD, K = refdi.model_matrices(m, f)
    Ui = zon.U.interval

    def loss_on_v(v, xb1):
        return cp.norm(xb1[0, :], p=2) ** 2 + 1e-2 * cp.norm(v[0], p=1) + 1e-2 * cp.norm(v[1], p=1)
    return build_parametric_qp(Ah, Bh, CK, DK, DD, K, zon.W.center, zon.W.generators, refdi.X_LOW, refdi.X_HIGH, Ui.left_limit, Ui.right_limit,
                               2, loss_on_v if penalised else common.loss_di, common.nocons, 2 if penalised else None)


@pytest.mark.parametrize("penalised", [False, True])
def test_product_builder_reproduces_the_reference_inputs(fig, penalised):
    """The product's condensed QP (tzddpc_amd.builder) solved by the oracle's solver: the same three checks, then the closed loop of
    examples/1.double_integrator_sim.py:75-90 on the reference's disturbances."""
    g, m, f = fig
    qp = _product_qp(m, f, penalised)

    def solve(xb, e):
        sol = common.oracle_solution(qp, xb, e, tol=1e-11)
        assert sol["status"] == "solved"
        return sol["v"][0], sol["xbar"][1]
    _check_steps(solve, m, f, penalised)
    x = g["x"][0].copy(); xbar = x.copy(); e = np.zeros(2)
    for t in range(12):
        v0, xb1 = solve(xbar, e)
        u = f["K"] @ e + v0                                                           # :86
        x = refdi.A_TRUE @ x + refdi.B_TRUE * u + m["w"][t]                           # :87
        xbar = xb1; e = x - xbar                                                      # :85, :89
        tol = refdi.TOL_LOOP if penalised else (refdi.TOL_ACTIVE if t < 4 else 0.05)
        assert np.abs(x - g["x"][t + 1]).max() <= tol, (t, x, g["x"][t + 1])


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_ze1_polygons_match_the_drawn_ones(fig, which):
    """`Ze[1]` as the reference exports and draws it (tzddpc/tzddpc.py:377; examples/1.double_integrator_sim.py:82-83, :165-167):
    MdataK <e, 0> + Mdelta <[xbar; v], 0> + W, moved by xbar_{t+1}, reduced to order <= 3, as a polygon -- vertex for vertex."""
    g, m, f = fig
    Ah, Bh, CK, DK, DD, K = refdi.model_matrices(m, f)
    if which == "oracle":
        from oracle.zonolite import Zonotope
        _, s, idn = _oracle_model(m, f)
        MK, MD, W = idn["MdataK"], idn["Mdelta"], s["W"]
    else:
        from tzddpc_amd.zonotope import Zonotope, boxed_matrix_zonotope
        MK, MD = boxed_matrix_zonotope(CK, DK), boxed_matrix_zonotope(np.zeros((2, 3)), DD)
        W = Zonotope(np.zeros(2), refdi.W_GEN)
    for t in range(12):
        Ze1 = MK * Zonotope(f["e"][t], np.zeros((2, 1))) + (MD * Zonotope(np.r_[f["xbar"][t], f["v"][t]], np.zeros((3, 1))) + W)
        Z = Ze1 + f["xbar"][t + 1]
        order = np.asarray(Z.generators).shape[1] / 2.0                                # pyzonotope's Zonotope.order: generators / dimension
        Z = Z.reduce(min(3, int(order)))
        if which == "product":
            V = Z.polygon_vertices()
        else:
            V = Z.compute_vertices()
        V = np.asarray(V, float)
        ctr = V.mean(axis=0)
        V = V[np.argsort(np.arctan2(V[:, 1] - ctr[1], V[:, 0] - ctr[0]))]
        # collinear vertices (two axis-aligned generators in a row) drop out of the drawn outline
        keep = []
        for i in range(len(V)):
            a, b, c_ = V[i - 1], V[i], V[(i + 1) % len(V)]
            if abs((b[0] - a[0]) * (c_[1] - b[1]) - (b[1] - a[1]) * (c_[0] - b[0])) > 1e-12:
                keep.append(b)
        V = np.array(keep)
        R = g["polygons"][t + 1]
        rc = R.mean(axis=0)
        R = R[np.argsort(np.arctan2(R[:, 1] - rc[1], R[:, 0] - rc[0]))]
        assert V.shape == R.shape, (t, V.shape)
        np.testing.assert_allclose(V, R, atol=5e-8)
