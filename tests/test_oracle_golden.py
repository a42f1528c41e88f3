"""-m "not gpu": the committed golden vectors (provenance: oracle only, tests/golden/make_golden.py) against
  * the oracle itself (they regenerate), * the product's identification on the same data sets, * the product's builder
(its QP solved by the oracle's solver must have the golden optimum)."""
import os

import numpy as np
import pytest

from tests import common

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = ["di_n2", "di_sim_n5", "di_n5", "di_n20", "di_n20_k1", "di_n20_k2", "pulley_n10", "dim5_n20", "dim5m2_n20", "dim5m2q_n20", "di2in_n10", "di2in_n10_k1"]


def _make_golden():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("case", ALL)
def test_goldens_regenerate_from_the_oracle(case):
    mg = _make_golden()
    g = np.load(os.path.join(GOLD, f"{case}.npz"))
    assert str(g["provenance"]).startswith("oracle only")
    sysname, loss, cons, N, k0 = mg.CASES[case]
    s, u, x, idn = mg.identified(sysname)
    np.testing.assert_array_equal(x, g["data_x"]); np.testing.assert_array_equal(u, g["data_u"])      # fixtures regenerate bit-for-bit
    np.testing.assert_allclose(idn["K"], g["K"], rtol=1e-12)
    b = 1                                                    # one of the four points per case keeps the CPU suite short
    sol = mg.solve_point(s, idn, N, k0, loss, cons, g["x0"][b], g["e0"][b])
    assert abs(sol["cost"] - g["cost"][b]) <= 1e-9 * (1 + abs(g["cost"][b]))
    np.testing.assert_allclose(sol["v"][0], g["v"][b, 0], atol=1e-8)
    np.testing.assert_allclose(sol["e0_rx"], g["e0_rx"][b], atol=1e-14)
    assert g["cert"].max() < 1e-9


@pytest.mark.parametrize("sysname", ["di_sim", "di_cc", "pulley", "dim5_w001", "dim5m2_w001", "di2in"])
def test_product_identification_equals_oracle(sysname):
    """Mdata, MdataK, Mdelta after reduce(1) (reference tzddpc/tzddpc.py:81-83, 119-128) of the product and of
    oracle.harness.identify on the four benchmark data sets: centres, generators and the boxed magnitudes."""
    from oracle import harness as H
    from oracle.collapsed import single_entry_abs
    from tzddpc_amd import TZDDPC, Data, Theta
    from tzddpc_amd.harness import system
    s = H.system(sysname)
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], np.random.default_rng(25))
    idn = H.identify(u, x, s["W"])
    A, B, zon, T = system(sysname)
    ctl = TZDDPC.__new__(TZDDPC); ctl.device = 0; ctl._native = None; ctl.qp = None
    ctl.update_identification_data(Data(u, x))
    ctl.build_zonotopes_theta(zon, theta=Theta(idn["K"], np.zeros_like(A), np.zeros_like(B)))
    for name in ("Mdata", "MdataK", "Mdelta"):
        mine, ref = getattr(ctl, name), idn[name]
        np.testing.assert_allclose(mine.center, ref.center, rtol=0, atol=1e-13)
        assert len(mine.generators) == ref.num_generators
        np.testing.assert_allclose(np.abs(np.asarray(mine.generators)).sum(axis=0), np.abs(ref.generators).sum(axis=0), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(ctl.MdataK.single_entry_magnitudes(), single_entry_abs(idn["MdataK"]), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(ctl.Mdelta.single_entry_magnitudes(), single_entry_abs(idn["Mdelta"]), rtol=1e-12, atol=1e-15)
    # the product's default gain (LQR on the identified centre) is the oracle's stand-in gain
    ctl2 = TZDDPC.__new__(TZDDPC); ctl2.device = 0; ctl2._native = None; ctl2.qp = None
    ctl2.update_identification_data(Data(u, x))
    ctl2.build_zonotopes_theta(zon)
    np.testing.assert_allclose(ctl2.theta.K, idn["K"], rtol=1e-10)


@pytest.mark.parametrize("case", ALL)
def test_product_builder_has_the_golden_optimum(case):
    """The product's (condensed, grouped-epigraph) QP, assembled from the golden data set and gain and solved by the oracle's
    solver, has the optimum the oracle found on its own (uncondensed, component-epigraph) statement of the problem."""
    g = np.load(os.path.join(GOLD, f"{case}.npz"))
    qp = common.qp_from_golden(case, g)
    rows = common.golden_tube_rows(qp)
    for b in (0, 2):
        ref = common.oracle_solution(qp, g["x0"][b], g["e0"][b], tol=1e-12)
        assert ref["status"] == "solved" and max(ref["cert"]["primal"], ref["cert"]["dual"], ref["cert"]["comp"]) < 1e-9
        assert abs(ref["cost"] - g["cost"][b]) <= 1e-8 * (1 + abs(g["cost"][b]))
        if case in common.NONUNIQUE:                         # optimal face: the objective and the priced coordinate are what is defined
            np.testing.assert_allclose(ref["xbar"][:qp.N, 1], g["xbar"][b, :qp.N, 1], atol=1e-6 * (1 + np.abs(g["xbar"][b]).max()))
            continue
        np.testing.assert_allclose(ref["v"][0], g["v"][b, 0], atol=1e-6 * (1 + np.abs(g["v"][b]).max()))
        np.testing.assert_allclose(ref["xbar"][1], g["xbar"][b, 1], atol=1e-6 * (1 + np.abs(g["xbar"][b]).max()))
        # active tube rows agree wherever the golden complementarity is clear-cut
        for (k, c, side), r in rows.items():
            sl, y = g["slack"][b, k, c, side], g["y"][b, k, c, side]
            if max(sl, y) > 1e-8 and (sl < 1e-2 * y or y < 1e-2 * sl):
                assert bool(ref["active"][r]) == bool(g["active"][b, k, c, side]), (k, c, side)


@pytest.mark.parametrize("name,sysname", [("di_n2_closed_loop", "di_sim"), ("di2in_n5_closed_loop", "di2in"), ("pulley_n4_closed_loop", "pulley")])
def test_closed_loop_golden_is_consistent(name, sysname):
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    from oracle import harness as H
    s = H.system(sysname)
    A, B = s["A"], s["B"]
    x, u, noise = g["x"], g["u"], g["noise"]
    # plant recursion of examples/1.double_integrator_sim.py:85 holds along the stored trajectory
    for b in range(x.shape[0]):
        for t in range(u.shape[1]):
            np.testing.assert_allclose(x[b, t + 1], A @ x[b, t] + (B @ u[b, t]) + noise[b, t], atol=1e-12)
    Xi = s["X"].interval
    assert np.all(x >= Xi.left_limit - 1e-9) and np.all(x <= Xi.right_limit + 1e-9)      # robust constraint satisfaction
    Ui = s["U"].interval
    assert np.all(u >= Ui.left_limit - 1e-9) and np.all(u <= Ui.right_limit + 1e-9)


def reference_pulley_envelope_checks(x, g, label):
    """x: (B, 201, 4) closed-loop states of the pulley example's loop (N = 2, x0 = 0, 200 steps, noise c + G U(-1, 1)) against the
    statistics of the reference's stored runs (tests/golden/pulley_reference_stats.npz, extracted from
    examples/results/pulley.xtzddpc.npy).  The reference's runs are un-seeded (noise AND data set), so this is an envelope, not a
    trajectory comparison: it is the only link to numbers the reference itself produced."""
    assert x.shape[1:] == (201, 4) and tuple(g["shape"][1:]) == (201, 4)
    np.testing.assert_array_equal(x[:, 0], 0.0)
    # reference: 0.916 ... 1.101: the nominal state reaches the target in ONE step, the plant lands at 1 + w + (model mismatch of
    # the identified B times v0, a few 1e-3), |w| <= 0.1
    first = x[:, 1, 0]
    assert first.min() >= 0.90 - 5e-3 and first.max() <= 1.10 + 5e-3, (label, first.min(), first.max())
    assert abs(first.mean() - 1.0) <= 0.005 + 4 * 0.0578 / np.sqrt(x.shape[0]), (label, first.mean())        # 4 sigma of the mean of B draws of 0.1 U(-1, 1)
    assert g["first_step"][:, 0].min() >= 0.90 and g["first_step"][:, 0].max() <= 1.11          # ... in the reference's runs as well
    tail = x[:, -50:]
    dm = np.abs(tail.mean(axis=(0, 1)) - g["tail_mean"])
    assert np.all(dm <= 0.01 + 3 * g["tail_std"] / np.sqrt(50 * 5)), (label, dm)              # 1.000 +- 0.01 on state 0
    assert abs(tail[:, :, 0].mean() - 1.0) <= 0.01, (label, tail[:, :, 0].mean())
    ds = np.abs(tail.std(axis=(0, 1)) - g["tail_std"])
    assert np.all(ds <= 0.01 + 0.1 * g["tail_std"]), (label, ds)                               # 0.058 = std of 0.1 U(-1, 1) on state 0
    # global envelope.  The plant is a shift register (companion form of examples/2.pulley_sim.py:39-43) driven by ONE noise generator
    # 0.1 * ones(4): state k is state 0 delayed by k steps plus k further noise draws, so once state 0 tracks 1 + w its support is
    # [-0.1 k, 1 + 0.1 (k + 1)] (the lower end is reached while the register fills).  The reference's five runs lie inside (min
    # 0 / -0.085 / -0.085 / -0.096, max 1.10 / 1.19 / 1.27 / 1.32); so must every one of our trajectories, and from 256 x 200 draws
    # they come closer to the ends than 5 x 200 do.
    k = np.arange(4)
    lo, hi = -0.1 * k - 0.05, 1.0 + 0.1 * (k + 1) + 0.05          # + 0.05: the error feedback K e does not vanish in one step
    assert np.all(g["state_min"] >= lo) and np.all(g["state_max"] <= hi)
    assert np.all(x >= lo) and np.all(x <= hi), (label, x.min(axis=(0, 1)), x.max(axis=(0, 1)))
    # after the shift register has filled (4 steps) every run of the reference stays inside our per-state range (+ 0.05)
    assert np.all(g["step_min"][4:] >= x[:, 4:].min(axis=(0, 1)) - 0.05) and np.all(g["step_max"][4:] <= x[:, 4:].max(axis=(0, 1)) + 0.05)


def test_oracle_closed_loop_lives_in_the_reference_pulley_envelope():
    """The ORACLE alone (oracle.harness data + identification, oracle.collapsed problem, oracle.qp_ipm solver, cold start every
    step) runs the loop of reference examples/2.pulley_sim.py:62-103 and lands in the envelope of the reference's stored runs."""
    mg = _make_golden()
    from oracle import harness as H
    g = np.load(os.path.join(GOLD, "pulley_reference_stats.npz"))
    s, u, x, idn = mg.identified("pulley")
    Bn, T = 6, 200
    rng = np.random.default_rng(2024)
    Wc, WG = np.asarray(s["W"].center, float), np.asarray(s["W"].generators, float)
    noise = Wc[None, None] + np.einsum("ig,btg->bti", WG, rng.uniform(-1.0, 1.0, size=(Bn, T, WG.shape[1])))    # W.sample(): c + G U(-1, 1)
    xs, us, cost = mg.closed_loop(s, idn, 2, None, H.loss_pulley, None, np.zeros((Bn, 4)), noise)
    reference_pulley_envelope_checks(xs, g, "oracle")
    Ui = s["U"].interval
    assert np.all(us >= Ui.left_limit - 1e-8) and np.all(us <= Ui.right_limit + 1e-8)
