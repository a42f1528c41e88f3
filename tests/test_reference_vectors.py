"""-m "not gpu": the reference's own stored closed-loop runs (tests/golden/pulley_reference_vectors.npz, see tests/refpulley.py)
replayed by the ORACLE and by the PRODUCT'S BUILDER (its QP solved by the oracle's solver) -- the device replays them in
tests/test_gpu_parity.py::test_device_replays_the_reference_pulley_runs."""
import os

import numpy as np
import pytest

from tests import common, refpulley

GOLD = refpulley.GOLD


def _make_golden():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def test_reference_vectors_are_what_the_stored_runs_say():
    """The fixture itself: plant recursion of examples/2.pulley_sim.py:91 to rounding, disturbances inside W, inputs inside U, states
    inside X, and the affine law u_t = K x_t + g* holding to 1e-12 from step 30 on in every run (all tube rows inactive)."""
    from oracle import harness as H
    g = refpulley.vectors()
    s = H.system("pulley")
    A, B = s["A"], s["B"]
    assert g["x"].shape == (6, 201, 4) and g["u"].shape == (6, 200) and g["w"].shape == (6, 200)
    for r in range(refpulley.N_RUNS):
        x, u, w = g["x"][r], g["u"][r], g["w"][r]
        np.testing.assert_allclose(x[1:], x[:-1] @ A.T + np.outer(u, B[:, 0]) + np.outer(w, np.ones(4)), rtol=0, atol=1e-14)
        assert np.abs(w).max() <= 0.1                                            # W = <0, 0.1 ones(4)>
        c, d, res = refpulley.affine_law(x, u)
        assert res <= 1e-12
        np.testing.assert_allclose(c, g["K"][r], rtol=0, atol=1e-10)
        assert abs(d - g["g_star"][r]) <= 1e-10
        assert np.abs(g["g"][r][13:] - g["g_star"][r]).max() <= 1e-12           # converged offsets
        assert abs(u[0] - g["g"][r][0]) == 0.0                                   # e_0 = 0: u_0 = v0_0
    Xi, Ui = s["X"].interval, s["U"].interval
    assert np.all(g["x"] >= Xi.left_limit) and np.all(g["x"] <= Xi.right_limit)
    assert np.all(g["u"] >= Ui.left_limit[0]) and np.all(g["u"] <= Ui.right_limit[0])
    # the runs' gains: within 6e-3 of the dead-beat row -A[0, :] of the true plant, 9e-3 apart from each other (data-set spread)
    assert np.abs(g["K"] + A[0][None]).max() <= 6e-3


def _oracle_identified(K):
    mg = _make_golden()
    from oracle import harness as H
    s = H.system("pulley")
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], np.random.default_rng(25))
    return mg, H, s, H.identify(u, x, s["W"], K=np.atleast_2d(K))


@pytest.mark.parametrize("run", range(refpulley.N_RUNS))
def test_oracle_replays_reference_run_with_own_data_set(run):
    """Comparison (A) of tests/refpulley.py: reference disturbances + reference gain + OUR data set -> the reference's states and
    inputs within the data-set spread the reference's own six runs show."""
    g = refpulley.vectors()
    mg, H, s, idn = _oracle_identified(g["K"][run])
    xs, us, _ = mg.closed_loop(s, idn, 2, None, H.loss_pulley, None, np.zeros((1, 4)), refpulley.noise_of(g, run))
    bx, bu = refpulley.spread_bounds(g, run)
    assert np.abs(xs[0] - g["x"][run]).max() <= bx, (np.abs(xs[0] - g["x"][run]).max(), bx)
    assert np.abs(us[0, :, 0] - g["u"][run]).max() <= bu
    c, d, res = refpulley.affine_law(xs[0], us[0, :, 0])
    assert res <= 1e-7                                                            # the same law as the reference's run ...
    np.testing.assert_allclose(c, g["K"][run], rtol=0, atol=1e-6)
    assert g["g_star"].min() - 3e-3 <= d <= g["g_star"].max() + 3e-3             # ... with an offset inside the reference's range


@pytest.mark.parametrize("run", range(refpulley.N_RUNS))
def test_oracle_replays_reference_run_with_admissible_model(run):
    """Comparison (B): with a centre inside OUR Mdata that reproduces the run's nine informative offsets the oracle reproduces all 200
    stored states and inputs of the run to 2e-7."""
    g = refpulley.vectors()
    K = g["K"][run]
    mg, H, s, idn = _oracle_identified(K)
    M0 = np.hstack([idn["A"], idn["B"]])
    M, res = refpulley.fit_admissible_model(M0, K, g["g"][run])
    assert res <= 1e-7
    box = np.abs(idn["Mdata"].generators).sum(axis=0)                             # interval hull of our matrix zonotope
    assert np.all(np.abs(M - M0) <= 0.3 * box), (np.abs(M - M0) / box).max()
    idn["A"], idn["B"] = M[:, :4], M[:, 4:]
    idn["MdataK"].center = M[:, :4] + M[:, 4:] @ np.atleast_2d(K)
    xs, us, _ = mg.closed_loop(s, idn, 2, None, H.loss_pulley, None, np.zeros((1, 4)), refpulley.noise_of(g, run))
    assert np.abs(xs[0] - g["x"][run]).max() <= refpulley.TOL_ADMISSIBLE
    assert np.abs(us[0, :, 0] - g["u"][run]).max() <= refpulley.TOL_ADMISSIBLE


def product_controller(K, M=None, device_build=False):
    """Product controller for the pulley loop of examples/2.pulley_sim.py (N = 2) with gain K; `M` replaces the identified centre
    [Ahat | Bhat] (and with it the centre of MdataK).  Host-only unless device_build."""
    from tzddpc_amd import TZDDPC, Theta
    from tzddpc_amd.harness import generate_trajectories, system
    A, B, zon, T = system("pulley")
    data = generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(25))
    if device_build:
        ctl = TZDDPC(data)
    else:
        ctl = TZDDPC.__new__(TZDDPC); ctl.device = 0; ctl._native = None; ctl.qp = None
        ctl.update_identification_data(data)
    K = np.atleast_2d(K)
    ctl.build_zonotopes_theta(zon, theta=Theta(K, np.zeros_like(A), np.zeros_like(B)))
    if M is not None:
        ctl.Mdata.center = np.array(M, float)
        ctl.MdataK.center = M[:, :4] + M[:, 4:] @ K
    return ctl, (A, B, zon)


@pytest.mark.parametrize("run", [0, 3, 5])
def test_product_builder_replays_reference_run(run):
    """The product's condensed QP of the pulley loop (tzddpc_amd.builder), solved step by step by the oracle's solver, in the loop
    of examples/2.pulley_sim.py:80-94 on the run's disturbances: comparison (B) for the product's formulation, no GPU involved."""
    from tzddpc_amd.builder import build_parametric_qp
    g = refpulley.vectors()
    K = g["K"][run]
    ctl0, _ = product_controller(K)
    M0 = ctl0.Mdata.center.copy()
    M, _ = refpulley.fit_admissible_model(M0, K, g["g"][run])
    ctl, (A, B, zon) = product_controller(K, M)
    Xi, Ui = zon.X.interval, zon.U.interval
    qp = build_parametric_qp(ctl.Mdata.center[:, :4], ctl.Mdata.center[:, 4:], ctl.MdataK.center, ctl.MdataK.single_entry_magnitudes(),
                             ctl.Mdelta.single_entry_magnitudes(), ctl.theta.K, zon.W.center, zon.W.generators, Xi.left_limit, Xi.right_limit,
                             Ui.left_limit, Ui.right_limit, 2, common.loss_pulley, common.nocons, None)
    x = np.zeros(4); xbar = np.zeros(4); e = np.zeros(4)
    w = g["w"][run]
    dx = du = 0.0
    for t in range(200):
        sol = common.oracle_solution(qp, xbar, e, tol=1e-11)
        assert sol["status"] == "solved"
        u = ctl.theta.K @ e + sol["v"][0]                                         # examples/2.pulley_sim.py:90
        x = A @ x + B @ u + w[t]                                                  # :91
        xbar = sol["xbar"][1]; e = x - xbar                                       # :89, :93
        dx = max(dx, np.abs(x - g["x"][run][t + 1]).max()); du = max(du, abs(u[0] - g["u"][run][t]))
    assert dx <= refpulley.TOL_ADMISSIBLE and du <= refpulley.TOL_ADMISSIBLE, (dx, du)
