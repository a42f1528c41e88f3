"""literal generator stacking == collapsed form (oracle side), and product builder == oracle, all on CPU.

This is what pins the N = 20 formulation: the reference's literal algebra (reference tzddpc/tzddpc.py:172-207, :283-324) is
only runnable for small N (generator counts 24, 64, 343, ...), so the identity is checked there to 1e-12 and the collapsed
form is what is solved at N = 20.
"""
import numpy as np
import pytest

from oracle import collapsed as C, harness as H, literal as L
from oracle.qp_ipm import solve_qp
from tests import common


SYS = [("di_sim", H.loss_di, None), ("pulley", H.loss_pulley, None), ("dim5", H.loss_dim5, H.constraints_dim5)]


@pytest.mark.parametrize("name,loss,cons", SYS)
@pytest.mark.parametrize("N,k0", [(2, None), (3, None), (3, 1), (4, 2)])
def test_literal_equals_collapsed(name, loss, cons, N, k0):
    if name == "dim5" and N >= 4:
        pytest.skip("literal stacking of the 5-dim system at N=4 has 1e5 generators (minutes)")
    s = H.system(name)
    rng = np.random.default_rng(25)
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], rng)
    idn = H.identify(u, x, s["W"])
    n = s["B"].shape[0]
    x0 = s["X0"].center + 0.01
    e0 = 0.02 * rng.standard_normal(n)
    args = (idn["A"], idn["B"], idn["MdataK"], idn["Mdelta"], idn["K"], s["W"], s["X"], s["U"], N, e0, x0, loss, cons, k0)
    lp = L.build_literal(*args)
    cq = C.build_collapsed(*args)
    for _ in range(4):                                   # interval hulls agree at random decision vectors
        xi = rng.standard_normal(lp.nxi)
        cr = C.collapsed_radii(cq, xi)
        for Zl, (cc, rx, ru) in zip(lp.Ze, cr):
            Zn = Zl.value(xi)
            np.testing.assert_allclose(Zn.center, cc, atol=1e-12)
            np.testing.assert_allclose(np.abs(Zn.generators).sum(axis=1), rx, atol=1e-12)
            np.testing.assert_allclose(np.abs(idn["K"] @ Zn.generators).sum(axis=1), ru, atol=1e-12)
    ql = L.to_qp(lp)
    r1 = solve_qp(ql["P"], ql["q"], ql["A"], ql["l"], ql["u"])
    r2 = solve_qp(cq["P"], cq["q"], cq["A"], cq["l"], cq["u"])
    if r1.status != "solved":                            # e.g. W = 0.1 tubes that do not fit: both must agree on that too
        assert r2.status != "solved"
        return
    assert max(r1.cert["primal"], r1.cert["dual"], r1.cert["comp"]) < 1e-8
    assert max(r2.cert["primal"], r2.cert["dual"], r2.cert["comp"]) < 1e-8
    o1, o2 = r1.obj + ql["r"], r2.obj + cq["r"]
    assert abs(o1 - o2) <= 1e-8 * (1 + abs(o1))
    v1 = r1.x[(N + 1) * n:(N + 1) * n + s["B"].shape[1]]
    v2, _ = C.extract(cq, r2.x)
    np.testing.assert_allclose(v1, v2[0], atol=1e-6)     # consumed output v[0]


@pytest.mark.parametrize("case", ["di_n5", "di_n20", "di_n20_k1", "pulley_n10", "dim5_n20"])
def test_product_builder_matches_oracle_collapsed(case):
    sysname, _, _, N, k0 = common.CASES[case]
    oloss = H.loss_dim5 if sysname.startswith("dim5") else (H.loss_pulley if sysname == "pulley" else H.loss_di)
    ocons = H.constraints_dim5 if sysname.startswith("dim5") else None
    ctl, qp, (A, B, zon) = common.identified_qp(case)
    s = H.system(sysname)
    from oracle.zonolite import MatrixZonotope, Zonotope
    MdataK = MatrixZonotope(ctl.MdataK.center, ctl.MdataK.generators)
    Mdelta = MatrixZonotope(ctl.Mdelta.center, ctl.Mdelta.generators)
    n = qp.n
    x0s, e0s = common.sample_params(zon, n, 2)
    for x0, e0 in zip(x0s, e0s):
        cq = C.build_collapsed(ctl.Mdata.center[:, :n], ctl.Mdata.center[:, n:], MdataK, Mdelta, ctl.theta.K,
                               Zonotope(zon.W.center, zon.W.generators), Zonotope(zon.X.center, zon.X.generators),
                               Zonotope(zon.U.center, zon.U.generators), N, e0, x0, oloss, ocons, k0)
        r = solve_qp(cq["P"], cq["q"], cq["A"], cq["l"], cq["u"], tol=1e-11)
        ref = common.oracle_solution(qp, x0, e0, tol=1e-11)
        assert r.status == "solved" and ref["status"] == "solved"
        assert max(ref["cert"]["primal"], ref["cert"]["dual"], ref["cert"]["comp"]) < 1e-8
        o = r.obj + cq["r"]
        assert abs(o - ref["cost"]) <= 1e-7 * (1 + abs(o))
        v, xb = C.extract(cq, r.x)
        np.testing.assert_allclose(ref["v"][0], v[0], atol=2e-6)
        np.testing.assert_allclose(ref["xbar"][1], xb[1], atol=2e-6)


@pytest.mark.parametrize("wc", [None, (3e-4, -2e-4)])
@pytest.mark.parametrize("N,k0", [(3, None), (4, 1), (2, None)])
def test_product_literal_problem_matches_oracle_literal_for_dense_generators(N, k0, wc):
    """Dense matrix-zonotope generators (Girard order 2): the product's literal problem (`build_parametric_qp(..., literal=stack)`,
    epigraph variables only for the decision-dependent generator entries, the rest numeric through theta) has the optimum of the
    oracle's literal restatement (every generator entry an epigraph, reference tzddpc/tzddpc.py:172-207 / :283-324).
    `wc`: a disturbance zonotope with a non-zero centre -- the constant part of the tube centres then is not zero and must enter
    the rows exactly once (round-2 advisor finding: it was subtracted in the rows AND carried by theta)."""
    from oracle import harness as H, literal as OL
    from oracle.qp_ipm import solve_qp
    from oracle.zonolite import Zonotope as OZ
    from tests import common
    from tzddpc_amd.builder import build_parametric_qp, theta_reference
    from tzddpc_amd.genstack import build_stack, count_generators
    from tzddpc_amd.zonotope import MatrixZonotope as PMZ, Zonotope as PZ
    s = H.system("di_cc"); rng = np.random.default_rng(25)
    if wc is not None:
        s = dict(s); s["W"] = OZ(np.asarray(wc), np.asarray(s["W"].generators))
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], rng)
    idn = H.identify(u, x, s["W"])
    dK, dD = idn["MdataK_raw"].reduce(2), idn["Mdelta_raw"].reduce(2)
    n, m = 2, 1
    pK, pD = PMZ(np.asarray(dK.center), np.asarray(dK.generators)), PMZ(np.asarray(dD.center), np.asarray(dD.generators))
    W = PZ(np.asarray(s["W"].center), np.asarray(s["W"].generators))
    st = build_stack(pK, pD, idn["K"], W, n, m, N, k0, nseg=N)
    tot, dec = count_generators(pK.num_generators, pD.num_generators, W.num_generators, N, k0, nseg=N)
    assert list(np.diff(st.seg_ptr)) == tot
    Xi, Ui = s["X"].interval, s["U"].interval
    qp = build_parametric_qp(idn["A"], idn["B"], np.asarray(dK.center), None, None, idn["K"], W.center, W.generators,
                             Xi.left_limit, Xi.right_limit, Ui.left_limit, Ui.right_limit, N, common.loss_di, common.nocons, k0, literal=st)
    assert qp.nz <= N * m + sum(dec) * (n + m) + 2 * N
    for b in range(3):
        x0 = np.asarray(s["X0"].center) + 0.05 * rng.standard_normal(2); e0 = 0.02 * rng.standard_normal(2)
        th = theta_reference(qp, x0, e0)
        r = solve_qp(qp.P, qp.q0 + qp.Qt @ th, qp.A, qp.l0 + qp.Lt @ th, qp.u0 + qp.Ut @ th, tol=1e-12)
        cost = r.obj + qp.r0 + qp.r1 @ x0 + x0 @ qp.R2 @ x0
        q = OL.to_qp(OL.build_literal(idn["A"], idn["B"], dK, dD, idn["K"], s["W"], s["X"], s["U"], N, e0, x0, H.loss_di, None, k0))
        ro = solve_qp(q["P"], q["q"], q["A"], q["l"], q["u"], tol=1e-12)
        assert r.status == "solved" == ro.status
        assert abs(cost - (ro.obj + q["r"])) <= 1e-7 * (1 + abs(cost))
        np.testing.assert_allclose(r.x[:m], ro.x[(N + 1) * n:(N + 1) * n + m], atol=1e-6)


def _cutting_plane_cpu(args, st, K, xl, xu, ul, uu, x0, e0, tol=1e-9, max_rounds=60):
    """The cutting-plane loop of `TZDDPC._solve_with_cuts` with the oracle's solver and numpy separation standing in for the device
    (same builder rows): returns (cost, v, xbar, rounds, number of patterns)."""
    from oracle.qp_ipm import solve_qp
    from tzddpc_amd.builder import build_parametric_qp, theta_reference
    n, m, N = st.n, st.m, st.N
    cuts = {}
    for rounds in range(max_rounds):
        qp = build_parametric_qp(*args, literal=st, cuts=cuts)
        th = theta_reference(qp, x0, e0)
        r = solve_qp(qp.P, qp.q0 + qp.Qt @ th, qp.A, qp.l0 + qp.Lt @ th, qp.u0 + qp.Ut @ th, tol=1e-12)
        assert r.status == "solved", r.status
        v = r.x[:N * m].reshape(N, m)
        xbar = (qp.Phi @ x0 + qp.Gam @ r.x[:N * m]).reshape(N + 1, n)
        zeta = np.concatenate([xbar[:N], v], axis=1)
        xi = np.zeros((N + 2, n + m)); xi[1, :n] = e0; xi[2:] = zeta
        val = st.m0 + np.einsum("gic,gc->gi", st.M, xi[st.src + 1])                 # every generator column
        added = 0
        for k in range(N):
            sl = slice(int(st.seg_ptr[k]), int(st.seg_ptr[k + 1]))
            c = st.c0[k] + st.cE[k] @ e0
            rx = np.abs(val[sl]).sum(axis=0); ru = np.abs(val[sl] @ K.T).sum(axis=0)
            cx = xbar[k] + c; cu = v[k] + K @ c
            for kind, comps, cen, rad, lo, hi, vals in (("x", n, cx, rx, xl, xu, val), ("u", m, cu, ru, ul, uu, val @ K.T)):
                for i in range(comps):
                    if max(cen[i] + rad[i] - hi[i], lo[i] - (cen[i] - rad[i])) > tol * (1 + max(abs(lo[i]), abs(hi[i]))):
                        fam = qp.families.get((k, kind, i))
                        if fam:
                            p = np.where(vals[np.asarray(fam), i] >= 0.0, 1.0, -1.0)
                            if p.tobytes() not in {q.tobytes() for q in cuts.get((k, kind, i), [])}:
                                cuts.setdefault((k, kind, i), []).append(p); added += 1
        if added == 0:
            cost = r.obj + qp.r0 + qp.r1 @ x0 + x0 @ qp.R2 @ x0
            return cost, v, xbar, rounds, sum(len(c) for c in cuts.values())
    raise AssertionError("cutting planes did not converge")


@pytest.mark.parametrize("N,k0", [(3, None), (4, 1), (5, 2)])
def test_cutting_plane_form_reaches_the_literal_optimum(N, k0):
    """Dense generators beyond the literal problem's size are solved in cutting-plane form (`build_parametric_qp(..., cuts=...)`:
    sign-pattern rows instead of one epigraph variable per generator entry).  On sizes where the oracle's literal restatement
    (reference tzddpc/tzddpc.py:172-207 / :283-324, every entry an epigraph) is still solvable the loop must end at ITS optimum."""
    from oracle import harness as H, literal as OL
    from oracle.qp_ipm import solve_qp
    from tests import common
    from tzddpc_amd.genstack import build_stack
    from tzddpc_amd.zonotope import MatrixZonotope as PMZ, Zonotope as PZ
    s = H.system("di_cc"); rng = np.random.default_rng(25)
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], rng)
    idn = H.identify(u, x, s["W"])
    dK, dD = idn["MdataK_raw"].reduce(2), idn["Mdelta_raw"].reduce(2)
    n, m = 2, 1
    pK, pD = PMZ(np.asarray(dK.center), np.asarray(dK.generators)), PMZ(np.asarray(dD.center), np.asarray(dD.generators))
    W = PZ(np.asarray(s["W"].center), np.asarray(s["W"].generators))
    st = build_stack(pK, pD, idn["K"], W, n, m, N, k0, nseg=N)
    Xi, Ui = s["X"].interval, s["U"].interval
    args = (idn["A"], idn["B"], np.asarray(dK.center), None, None, idn["K"], W.center, W.generators,
            Xi.left_limit, Xi.right_limit, Ui.left_limit, Ui.right_limit, N, common.loss_di, common.nocons, k0)
    K = np.atleast_2d(idn["K"])
    for b in range(3):
        x0 = np.asarray(s["X0"].center) + 0.05 * rng.standard_normal(2); e0 = 0.02 * rng.standard_normal(2)
        cost, v, xbar, rounds, ncut = _cutting_plane_cpu(args, st, K, np.asarray(Xi.left_limit), np.asarray(Xi.right_limit),
                                                         np.asarray(Ui.left_limit), np.asarray(Ui.right_limit), x0, e0)
        q = OL.to_qp(OL.build_literal(idn["A"], idn["B"], dK, dD, idn["K"], s["W"], s["X"], s["U"], N, e0, x0, H.loss_di, None, k0))
        ro = solve_qp(q["P"], q["q"], q["A"], q["l"], q["u"], tol=1e-12)
        assert ro.status == "solved"
        assert abs(cost - (ro.obj + q["r"])) <= 1e-7 * (1 + abs(cost)), (cost, ro.obj + q["r"], rounds, ncut)
        np.testing.assert_allclose(v[0], ro.x[(N + 1) * n:(N + 1) * n + m], atol=1e-6)
