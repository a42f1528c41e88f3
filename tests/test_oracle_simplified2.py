"""solve_simplified2 (reference tzddpc/tzddpc.py:381-500): the oracle's literal restatement against the product's condensed
parametric QP (CPU: both solved by the oracle's interior point; the device solve is compared in tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from oracle import harness as H
from oracle import simplified2 as S2
from oracle.zonolite import Zonotope as OZ
from tests import common

CASES2 = {
    # name: (system, oracle loss, product loss, oracle constraints, product constraints, N, sigma scale, delta scale)
    "di": ("di_sim", H.loss_di, common.loss_di, None, common.nocons, 5, 0.01, 0.01),
    "pulley": ("pulley", H.loss_pulley, common.loss_pulley, None, common.nocons, 6, 0.005, 0.002),
    "di2in": ("di2in", H.loss_di, common.loss_di, None, common.nocons, 5, 0.01, 0.01),          # two inputs
}


def problem_data(name, seed=25, line_x=False):
    sysname, lo, lp, co, cpn, N, ssig, sdel = CASES2[name]
    s = H.system(sysname)
    rng = np.random.default_rng(seed)
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], rng)
    idn = H.identify(u, x, s["W"])
    n, m = s["B"].shape
    if sysname == "pulley" and not line_x:     # the example's X has ONE generator (a segment): xbar[1:] in X (:422) is then infeasible
        s = dict(s); s["X"] = OZ(s["X"].center, 2.0 * np.eye(n))
    Zs = [OZ(ssig * rng.standard_normal(n) * 0.1, ssig * (np.eye(n) + 0.3 * rng.standard_normal((n, n)))) for _ in range(N)]
    dA = sdel * rng.standard_normal((n, n)); dB = sdel * rng.standard_normal((n, m))
    return dict(s=s, u=u, x=x, idn=idn, Zs=Zs, dA=dA, dB=dB, N=N, lo=lo, lp=lp, co=co, cp=cpn, n=n, m=m)


def params(d, B=3, seed=3):
    rng = np.random.default_rng(seed)
    n = d["n"]
    Xc = np.asarray(d["s"]["X"].center, float)
    x0 = Xc[None] + 0.2 * rng.standard_normal((B, n)) * np.abs(np.asarray(d["s"]["X"].generators)).sum(axis=1)[None] * 0.3
    e0 = 0.01 * rng.standard_normal((B, n))
    return x0, e0


@pytest.mark.parametrize("name", ["di", "pulley", "di2in"])
@pytest.mark.parametrize("ze_sum", ["radius", "columns"])
def test_condensed_problem_equals_literal_restatement(name, ze_sum):
    from oracle.qp_ipm import solve_qp
    from tzddpc_amd.builder import build_simplified2_qp, eliminate_equalities, theta_reference
    d = problem_data(name)
    s, idn = d["s"], d["idn"]
    K = idn["K"]
    qp = build_simplified2_qp(idn["A"] + idn["B"] @ K, idn["B"], K, d["dA"], d["dB"], s["W"].center, s["W"].generators,
                              [(Z.center, Z.generators) for Z in d["Zs"]], (s["X"].center, s["X"].generators),
                              (s["U"].center, s["U"].generators), d["N"], d["lp"], d["cp"], ze_sum)
    red, el = eliminate_equalities(qp)
    assert el is not None and red.nz >= d["N"] * d["m"]
    x0s, e0s = params(d)
    nv = d["N"] * d["m"]
    for b in range(x0s.shape[0]):
        o = S2.solve(idn["A"], idn["B"], K, d["dA"], d["dB"], s["W"], s["X"], s["U"], d["Zs"], d["N"], x0s[b], e0s[b], d["lo"], d["co"],
                     ze_sum=ze_sum)
        assert o["status"] == "solved", o["status"]
        th = theta_reference(qp, x0s[b], e0s[b])
        extra = 0.0 if qp.rt is None else float(qp.rt @ th)
        for prob, rec in ((qp, None), (red, el)):
            r = solve_qp(prob.P, prob.q0 + prob.Qt @ th, prob.A, prob.l0 + prob.Lt @ th, prob.u0 + prob.Ut @ th, tol=1e-12)
            # the unscaled eliminated problem of the two-input case stalls the numpy solver at a dual residual of 4e-9 (1e-12 asked):
            # a certificate below 1e-8 is accepted
            assert r.status == "solved" or max(r.cert["primal"], r.cert["dual"], r.cert["comp"]) < 1e-8, (r.status, r.cert)
            par = prob.f0 + prob.Ft @ th
            assert np.all(par >= prob.pl - 1e-9) and np.all(par <= prob.pu + 1e-9)
            x = r.x if rec is None else rec.x0 + rec.Xn @ x0s[b] + rec.Z @ r.x
            cost = r.obj + prob.r0 + prob.r1 @ x0s[b] + x0s[b] @ prob.R2 @ x0s[b] + extra
            assert abs(cost - o["result"]) <= 1e-7 * (1 + abs(o["result"])), (cost, o["result"])
            if name in ("di", "di2in"):                            # strictly convex in the trajectory: the optimiser is unique
                np.testing.assert_allclose(x[:nv].reshape(d["N"], d["m"]), o["v"], atol=1e-6 * (1 + np.abs(o["v"]).max()))


def test_literal_restatement_respects_its_own_constraints():
    """The oracle's solution satisfies the reference's constraint list when re-evaluated from scratch with zonotope algebra."""
    d = problem_data("di")
    s, idn = d["s"], d["idn"]
    K = idn["K"]; N = d["N"]
    x0s, e0s = params(d, 2)
    o = S2.solve(idn["A"], idn["B"], K, d["dA"], d["dB"], s["W"], s["X"], s["U"], d["Zs"], N, x0s[0], e0s[0], d["lo"], None)
    Acl = idn["A"] + idn["B"] @ K
    xbar, v, ubar = o["xbar"], o["v"], o["ubar"]
    np.testing.assert_allclose(xbar[0], x0s[0], atol=1e-10)
    Xi, Ui = s["X"].interval, s["U"].interval
    t2 = np.zeros(2); cen = e0s[0].copy(); G = np.zeros((2, 1)); t1 = None
    for k in range(N):
        np.testing.assert_allclose(xbar[k + 1], Acl @ xbar[k] + idn["B"] @ v[k], atol=1e-9)
        np.testing.assert_allclose(ubar[k], K @ xbar[k] + v[k], atol=1e-9)
        rad = np.abs(G).sum(axis=1); radu = np.abs(K @ G).sum(axis=1)
        assert np.all(xbar[k] + cen + rad <= Xi.right_limit + 1e-8) and np.all(xbar[k] + cen - rad >= Xi.left_limit - 1e-8)
        assert np.all(K @ cen + ubar[k] + radu <= Ui.right_limit + 1e-8) and np.all(K @ cen + ubar[k] - radu >= Ui.left_limit - 1e-8)
        assert np.all(np.abs(ubar[k] - s["U"].center) <= np.abs(s["U"].generators).sum(axis=1) + 1e-8)
        t1 = (s["W"] + d["Zs"][0]) if k == 0 else (t1 * Acl + (s["W"] + d["Zs"][k]))
        t2 = Acl.T @ t2 + d["dA"] @ xbar[k] + d["dB"] @ ubar[k]
        cen = np.linalg.matrix_power(Acl, k + 1) @ e0s[0] + t1.center + t2
        G = np.concatenate([np.zeros((2, 1)), t1.generators], axis=1)
        if k == 0:
            np.testing.assert_allclose(o["ze1"][:, 0], cen, atol=1e-9)
            np.testing.assert_allclose(o["ze1"][:, 1:], G, atol=0)


def test_segment_state_zonotope_is_reported_infeasible_by_both():
    """examples/2.pulley_sim.py:54: X = <1, 2 * ones(4, 1)> is a segment; the membership xbar[k] in X of :422 then pins all four
    states to one coefficient and the problem is infeasible -- the oracle says so, and the product's elimination turns the
    dependent equality rows into parameter tests that fail."""
    from tzddpc_amd.builder import build_simplified2_qp, eliminate_equalities, theta_reference
    d = problem_data("pulley", line_x=True)
    s, idn = d["s"], d["idn"]
    K = idn["K"]
    x0s, e0s = params(d, 1)
    o = S2.solve(idn["A"], idn["B"], K, d["dA"], d["dB"], s["W"], s["X"], s["U"], d["Zs"], d["N"], x0s[0], e0s[0], d["lo"], None)
    assert o["status"] != "solved" and not np.isfinite(o["result"])
    qp = build_simplified2_qp(idn["A"] + idn["B"] @ K, idn["B"], K, d["dA"], d["dB"], s["W"].center, s["W"].generators,
                              [(Z.center, Z.generators) for Z in d["Zs"]], (s["X"].center, s["X"].generators),
                              (s["U"].center, s["U"].generators), d["N"], d["lp"], d["cp"])
    red, el = eliminate_equalities(qp)
    par = red.f0 + red.Ft @ theta_reference(qp, x0s[0], e0s[0])
    assert np.any(par < red.pl) or np.any(par > red.pu)
