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
