"""-m gpu: the HIP path (through the C-ABI) against the oracle, the golden fixtures and domain properties."""
import os

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL = 1e-6          # north_star tolerance: state / input trajectories within 1e-6 relative


def _ctl_from_golden(case):
    from tzddpc_amd import TZDDPC, Data, Theta
    from tzddpc_amd.harness import system
    g = np.load(os.path.join(GOLD, f"{case}.npz"))
    sysname, loss, cons, N, k0 = common.CASES[case]
    A, B, zon, T = system(sysname)
    ctl = TZDDPC(Data(g["data_u"], g["data_x"]))
    ctl.build_zonotopes_theta(zon, theta=Theta(g["K"], np.zeros_like(A), np.zeros_like(B)))
    if k0 is None:
        ctl.build_problem(N, loss, cons)
    else:
        ctl.build_problem_simplified(k0, N, loss, cons)
    return ctl, g, (A, B, zon)


GOLDEN_CASES = ["di_n2", "di_sim_n5", "di_n5", "di_n20", "di_n20_k1", "di_n20_k2", "pulley_n10", "dim5_n20",
                "dim5m2_n20", "dim5m2q_n20", "di2in_n10", "di2in_n10_k1"]          # the last four: two inputs


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_golden_parity(built, case):
    """Device solve against the oracle-only goldens (tests/golden/make_golden.py): cost, consumed input / state, active tube rows."""
    ctl, g, _ = _ctl_from_golden(case)
    out = ctl.solve_batch(g["x0"], g["e0"], want_active=True)
    assert (out["status"] == 0).all(), out["status"]
    scale = 1.0 + np.abs(g["xbar"]).max()
    rows = common.golden_tube_rows(ctl.qp)
    for b in range(4):
        assert abs(out["cost"][b] - g["cost"][b]) <= 1e-7 * (1 + abs(g["cost"][b]))
        if case in common.NONUNIQUE:        # optimal face (two inputs, one priced state coordinate): objective and that coordinate
            np.testing.assert_allclose(out["xbar"][b, :ctl.qp.N, 1], g["xbar"][b, :ctl.qp.N, 1], atol=REL * scale)
            continue
        np.testing.assert_allclose(out["v"][b, 0], g["v"][b, 0], atol=REL * (1 + np.abs(g["v"][b]).max()))   # consumed input
        np.testing.assert_allclose(out["xbar"][b, 1], g["xbar"][b, 1], atol=REL * scale)                        # consumed state
        # active set: identical wherever the oracle's complementarity is not borderline
        act = out["active"][b].astype(bool)
        checked = 0
        for (k, c, side), r in rows.items():
            sl, y = g["slack"][b, k, c, side], g["y"][b, k, c, side]
            if max(sl, y) > 1e-8 and (sl < 1e-2 * y or y < 1e-2 * sl):
                assert act[r] == bool(g["active"][b, k, c, side]), (case, b, k, c, side)
                checked += 1
        assert checked > len(rows) // 2
    if case.startswith("di_") and common.CASES[case][4] is None:   # full problem: strictly convex in xbar_1..xbar_{N-1}
        N = common.CASES[case][3]
        np.testing.assert_allclose(out["xbar"][:, :N], g["xbar"][:, :N], atol=5e-6 * scale)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_tube_theta_matches_oracle(built, case):
    """K1 (tz_tube_kernel): the e0-dependent part of every tube -- centre C_K^p e0, radii of M_K^p <e0> and of K M_K^p <e0>
    (reference tzddpc/tzddpc.py:172-181, 191-192) -- as left in theta by the device against the oracle's collapsed recursion
    stored with the goldens, <= 1e-12 absolute."""
    ctl, g, _ = _ctl_from_golden(case)
    out = ctl.solve_batch(g["x0"], g["e0"])
    n, m, N = ctl.qp.n, ctl.qp.m, ctl.qp.N
    for b in range(4):
        th = ctl._native.debug_fetch(b, 0)
        assert th.size == 2 * n + N * (2 * n + m)
        np.testing.assert_array_equal(th[:n], g["x0"][b]); np.testing.assert_array_equal(th[n:2 * n], np.abs(g["x0"][b]))
        blk = th[2 * n:].reshape(N, 2 * n + m)
        np.testing.assert_allclose(blk[:, :n], g["e0_c"][b], rtol=0, atol=1e-12)
        np.testing.assert_allclose(blk[:, n:2 * n], g["e0_rx"][b], rtol=0, atol=1e-12)
        np.testing.assert_allclose(blk[:, 2 * n:], g["e0_ru"][b], rtol=0, atol=1e-12)


@pytest.mark.parametrize("sysname", ["di_sim", "pulley", "dim5_w001", "di2in", "dim5m2_w001"])
def test_tube_theta_matches_literal_generator_stacking(built, sysname):
    """The same device numbers against LITERAL stacking (oracle.zonolite: MatrixZonotope * Zonotope multiplies the generator
    count by gamma_K + 1 per product, exactly like the reference's :175,181) for N = 4 (powers 0..3) on all three systems."""
    from oracle import harness as H
    from oracle.zonolite import Zonotope as OZ
    from tzddpc_amd import TZDDPC, Data, Theta
    from tzddpc_amd.harness import system
    s = H.system(sysname)
    rng = np.random.default_rng(25)
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], rng)
    idn = H.identify(u, x, s["W"])
    A, B, zon, T = system(sysname)
    n, m = B.shape
    ctl = TZDDPC(Data(u, x))
    ctl.build_zonotopes_theta(zon, theta=Theta(idn["K"], np.zeros_like(A), np.zeros_like(B)))
    loss = {"di_sim": common.loss_di, "pulley": common.loss_pulley, "dim5_w001": common.loss_dim5, "di2in": common.loss_di, "dim5m2_w001": common.loss_dim5}[sysname]
    N = 4
    ctl.build_problem(N, loss, common.cons_dim5 if sysname.startswith("dim5") else common.nocons)
    Bn = 6
    x0 = np.tile(zon.X0.center, (Bn, 1)); e0 = 0.02 * rng.standard_normal((Bn, n))
    ctl.solve_batch(x0, e0)                                  # status irrelevant: theta is formed before the solve
    for b in range(Bn):
        blk = ctl._native.debug_fetch(b, 0)[2 * n:].reshape(N, 2 * n + m)
        Z = OZ(e0[b], np.zeros((n, 1)))
        for k in range(N):
            if k > 0:
                Z = idn["MdataK"] * Z                         # literal: [C Z, G_1 Z, ..., G_gamma Z]
            np.testing.assert_allclose(blk[k, :n], Z.center, rtol=0, atol=1e-12)
            np.testing.assert_allclose(blk[k, n:2 * n], np.abs(Z.generators).sum(axis=1), rtol=0, atol=1e-12)
            np.testing.assert_allclose(blk[k, 2 * n:], np.abs(idn["K"] @ Z.generators).sum(axis=1), rtol=0, atol=1e-12)


def test_full_size_against_c_oracle(built):
    """BASELINE config 2 size: 1024 trajectories, N = 20, every trajectory against the plain-C restatement."""
    from oracle.c_oracle import COracle
    ctl, (A, B, zon) = common.gpu_controller("di_n20")
    x0, e0 = common.sample_params(zon, 2, 1024, seed=3, e_scale=0.03, x_scale=0.3)
    out = ctl.solve_batch(x0, e0, want_active=True)
    ref = COracle(ctl.qp).solve_batch(x0, e0, threads=16, want_active=True)
    assert (out["status"] == 0).all() and (ref["status"] == 0).all()
    assert np.abs(out["cost"] - ref["cost"]).max() <= 1e-7 * (1 + np.abs(ref["cost"]).max())
    assert np.abs(out["v"][:, 0] - ref["v"][:, 0]).max() <= REL * (1 + np.abs(ref["v"]).max())
    assert np.abs(out["xbar"][:, 1] - ref["xbar"][:, 1]).max() <= REL * (1 + np.abs(ref["xbar"]).max())
    # north star: "identical active constraint sets".  active <=> slack < multiplier, both unscaled: two quantities of different
    # units, so on a row where BOTH are small (weakly active: strict complementarity nearly fails) two solvers that stop at
    # different points of the central path may fall on different sides.  Every row on which device and C oracle disagree is
    # looked up in a third, much tighter solve (numpy oracle, tol 1e-12) and must be such a degenerate row THERE: slack below 1 %
    # of the row scale AND multiplier below 1 % of the largest multiplier.  A row with a clear slack or a clear multiplier never flips.
    diff = out["active"] != ref["active"]
    flipped = np.argwhere(diff)
    worst_s = worst_y = 0.0
    for b in np.unique(flipped[:, 0]):
        o = common.oracle_solution(ctl.qp, x0[b], e0[b], tol=1e-12)
        assert o["status"] == "solved"
        ymax = np.abs(o["y"]).max()
        for r in flipped[flipped[:, 0] == b, 1]:
            sl, y = float(o["slack"][r]), float(abs(o["y"][r]))
            rs = 1.0 + np.abs(ctl.qp.A[r]).max()
            worst_s = max(worst_s, sl / rs); worst_y = max(worst_y, y / ymax)
            assert sl <= 1e-2 * rs and y <= 1e-2 * ymax, (int(b), int(r), ctl.qp.row_names[r], sl, y, ymax)
    print(f"active sets: {int(diff.sum())} of {diff.size} rows differ ({np.unique(flipped[:, 0]).size} of {diff.shape[0]} trajectories), all of them "
          f"weakly active in the tight solve (largest relative slack {worst_s:.1e}, largest relative multiplier {worst_y:.1e})")
    assert diff.mean() < 0.005
    # size-independent properties: the returned nominal trajectory obeys the data-center dynamics (reference :166-170)
    n = 2
    Ah, Bh = ctl.Mdata.center[:, :n], ctl.Mdata.center[:, n:]
    pred = np.einsum("ij,bkj->bki", Ah, out["xbar"][:, :-1]) + np.einsum("ij,bkj->bki", Bh, out["v"])
    np.testing.assert_allclose(out["xbar"][:, 1:], pred, atol=1e-9)
    np.testing.assert_allclose(out["xbar"][:, 0], x0, atol=0)


def test_determinism_and_batch_independence(built):
    ctl, (A, B, zon) = common.gpu_controller("di_n20")
    x0, e0 = common.sample_params(zon, 2, 257, seed=5)
    a = ctl.solve_batch(x0, e0)
    b = ctl.solve_batch(x0, e0)
    for k in ("v", "xbar", "cost"):
        assert np.array_equal(a[k], b[k])                       # same inputs -> bitwise same outputs
    idx = np.array([200, 3, 77, 256, 0])
    c = ctl.solve_batch(x0[idx], e0[idx])
    for k in ("v", "xbar", "cost"):
        assert np.array_equal(c[k], a[k][idx])                  # trajectories are independent units (shardable)
    # the multi-GPU partition (tzddpc_amd/dist.shard_range: contiguous, sizes differing by one): every shard solved on its own
    # equals its rows of the full batch bit for bit, stateless solves and closed loops alike -- for 2, 3 and 8 ranks
    from tzddpc_amd.dist import shard_range, vertex_noise
    noise = vertex_noise(zon.W.compute_vertices(), 0, 257, 8)
    xs = np.tile(zon.X0.center, (257, 1))
    full = ctl.simulate_batch(xs, noise, A, B)
    for ws in (2, 3, 8):
        for r in range(ws):
            lo, hi = shard_range(257, ws, r)
            part = ctl.solve_batch(x0[lo:hi], e0[lo:hi])
            for k in ("v", "xbar", "cost"):
                assert np.array_equal(part[k], a[k][lo:hi])
            if ws == 2 or r in (0, ws - 1):
                sim = ctl.simulate_batch(xs[lo:hi], noise[lo:hi], A, B)
                for k in ("x", "u", "cost"):
                    assert np.array_equal(sim[k], full[k][lo:hi])
    one = ctl.solve_batch(x0[:1], e0[:1])
    assert np.array_equal(one["v"], a["v"][:1])


def test_solve_api_matches_reference_surface(built):
    ctl, (A, B, zon) = common.gpu_controller("di_n2")
    xbar0 = zon.X0.center.copy(); e0 = np.array([0.01, -0.005])
    result, v, xbark, Zek = ctl.solve(xbar0, e0, verbose=False)
    assert isinstance(result, float) and v.shape == (2, 1) and xbark.shape == (3, 2)
    Z = Zek.Z.value                                             # examples/1.double_integrator_sim.py:89-90
    assert Z.shape == (2, 1 + 24)                               # Gamma_1 = 24 for the double integrator
    ref = common.oracle_solution(ctl.qp, xbar0, e0)
    assert abs(result - ref["cost"]) <= 1e-8 * (1 + abs(ref["cost"]))
    # interval hull of the literal Ze[1] equals the collapsed centre / radius the constraints of step 1 use (oracle side:
    # the same data set identified by the oracle, its collapsed tubes evaluated at the device's solution)
    from oracle import collapsed as OC, harness as H
    s = H.system("di_sim")
    d = ctl.dataset.original_data
    idn = H.identify(np.asarray(d.u, float), np.asarray(d.x, float), s["W"], K=ctl.theta.K)
    cq = OC.build_collapsed(idn["A"], idn["B"], idn["MdataK"], idn["Mdelta"], idn["K"], s["W"], s["X"], s["U"], 2, e0, xbar0, H.loss_di)
    c1, rx1, _ = OC.collapsed_radii(cq, np.concatenate([xbark.reshape(-1), v.reshape(-1)]))[1]
    np.testing.assert_allclose(Z[:, 0], c1, atol=1e-10)
    np.testing.assert_allclose(np.abs(Z[:, 1:]).sum(axis=1), rx1, atol=1e-10)
    with pytest.raises(Exception, match="unbounded"):           # reference :374-375 (also raised for infeasible)
        ctl.solve(np.array([50.0, 0.0]), e0)


@pytest.mark.parametrize("name,sysname,loss", [("di2in_n5_closed_loop", "di2in", "di"), ("pulley_n4_closed_loop", "pulley", "pulley")])
def test_longer_oracle_only_closed_loops(built, name, sysname, loss):
    """30- / 40-step closed loops solved step by step by the oracle alone (tests/golden/make_golden.py: oracle formulation, oracle
    solver, cold start every step) -- one of them with two inputs -- against the device's fused closed loop."""
    from tzddpc_amd import TZDDPC, Data, Theta
    from tzddpc_amd.harness import system
    gl = np.load(os.path.join(GOLD, f"{name}.npz"))
    A, B, zon, T = system(sysname)
    ctl = TZDDPC(Data(gl["data_u"], gl["data_x"]))
    ctl.build_zonotopes_theta(zon, theta=Theta(gl["K"], np.zeros_like(A), np.zeros_like(B)))
    ctl.build_problem(int(gl["horizon"]), {"di": common.loss_di, "pulley": common.loss_pulley}[loss], common.nocons)
    sim = ctl.simulate_batch(gl["x0"], gl["noise"], A, B)
    assert (sim["status"] == 0).all()
    np.testing.assert_allclose(sim["x"], gl["x"], atol=REL * (1 + np.abs(gl["x"]).max()))
    np.testing.assert_allclose(sim["u"], gl["u"], atol=REL * (1 + np.abs(gl["u"]).max()))
    np.testing.assert_allclose(sim["cost"], gl["cost"], rtol=1e-7, atol=1e-7)


def test_closed_loop_golden_and_c_oracle(built):
    from oracle.c_oracle import COracle
    ctl, g, (A, B, zon) = _ctl_from_golden("di_n2")
    gl = np.load(os.path.join(GOLD, "di_n2_closed_loop.npz"))
    sim = ctl.simulate_batch(gl["x0"], gl["noise"], A, B)
    assert (sim["status"] == 0).all()
    np.testing.assert_allclose(sim["x"], gl["x"], atol=REL * (1 + np.abs(gl["x"]).max()))
    np.testing.assert_allclose(sim["u"], gl["u"], atol=REL * 10)
    # longer horizon, more trajectories, against the plain-C closed loop
    from tzddpc_amd.dist import vertex_noise
    ctl2, (A, B, zon) = common.gpu_controller("di_n20")
    noise = vertex_noise(zon.W.compute_vertices(), 0, 64, 15)
    x0 = np.tile(zon.X0.center, (64, 1))
    s1 = ctl2.simulate_batch(x0, noise, A, B)
    s2 = common.c_oracle_for(ctl2).simulate_batch(x0, noise, A, B, threads=16)
    assert (s1["status"] == 0).all() and (s2["status"] == 0).all()
    np.testing.assert_allclose(s1["x"], s2["x"], atol=1e-6)
    np.testing.assert_allclose(s1["u"], s2["u"], atol=1e-6)
    # north-star bound: <= 1e-6 state error against a much tighter, cold-started solve of every step
    tight = COracle(ctl2.qp, warm_floor=0.0, tol=1e-11, mu_factor=1e-3, step_frac=0.99, max_iter=80).simulate_batch(x0, noise, A, B, threads=16)
    assert (tight["status"] == 0).all()
    np.testing.assert_allclose(s1["x"], tight["x"], atol=1e-6)
    np.testing.assert_allclose(s1["u"], tight["u"], atol=1e-6)
    Xi = zon.X.interval
    assert np.all(s1["x"] >= Xi.left_limit - 1e-9) and np.all(s1["x"] <= Xi.right_limit + 1e-9)


def test_device_pointer_path_equals_host_path(built):
    import torch
    ctl, (A, B, zon) = common.gpu_controller("di_n5")
    x0, e0 = common.sample_params(zon, 2, 33, seed=9)
    host = ctl.solve_batch(x0, e0)
    dev = torch.device("cuda", 0)
    tx, te = torch.from_numpy(x0).to(dev), torch.from_numpy(e0).to(dev)
    v = torch.empty((33, 5, 1), dtype=torch.float64, device=dev); xb = torch.empty((33, 6, 2), dtype=torch.float64, device=dev)
    cost = torch.empty(33, dtype=torch.float64, device=dev); st = torch.empty(33, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctl._native.solve_batch_ptr(33, tx.data_ptr(), te.data_ptr(), v.data_ptr(), xb.data_ptr(), cost.data_ptr(), st.data_ptr())
    ctl._native.sync()
    assert np.array_equal(v.cpu().numpy(), host["v"]) and np.array_equal(cost.cpu().numpy(), host["cost"])
    assert (st.cpu().numpy() == 0).all()


def test_warm_started_closed_loop_equals_cold(built):
    """tz_mpc_run / tz_simulate_batch warm-start the interior point from the previous step: same trajectories as cold starts."""
    import torch
    from tzddpc_amd.dist import vertex_noise
    ctl, (A, B, zon) = common.gpu_controller("di_n20")
    Bn, T = 96, 12
    noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
    x0 = np.tile(zon.X0.center, (Bn, 1))
    warm = ctl.simulate_batch(x0, noise, A, B)
    # cold reference: step by step through solve_batch (stateless, always cold) with the same plant recursion
    K = ctl.theta.K
    x = x0.copy(); xbar = x0.copy(); e = np.zeros_like(x0)
    xs = [x0.copy()]
    for t in range(T):
        out = ctl.solve_batch(xbar, e)
        assert (out["status"] == 0).all()
        u = e @ K.T + out["v"][:, 0]
        x = x @ A.T + u @ B.T + noise[:, t]
        xbar = out["xbar"][:, 1]; e = x - xbar
        xs.append(x.copy())
    xs = np.stack(xs, axis=1)
    assert (warm["status"] == 0).all()
    np.testing.assert_allclose(warm["x"], xs, atol=1e-6)
    # device-resident multi-step entry point
    dev = torch.device("cuda", 0)
    tx = torch.from_numpy(x0.copy()).to(dev); txb = tx.clone(); te = torch.zeros_like(tx)
    tw = torch.from_numpy(np.ascontiguousarray(noise.transpose(1, 0, 2))).to(dev)
    tA = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev); tB = torch.from_numpy(np.ascontiguousarray(B, dtype=np.float64)).to(dev)
    tu = torch.zeros((Bn, 1), dtype=torch.float64, device=dev); tc = torch.zeros(Bn, dtype=torch.float64, device=dev)
    ts = torch.zeros(Bn, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctl._native.mpc_run_ptr(Bn, T, tx.data_ptr(), txb.data_ptr(), te.data_ptr(), tw.data_ptr(), tA.data_ptr(), tB.data_ptr(),
                            tu.data_ptr(), tc.data_ptr(), ts.data_ptr())
    ctl._native.sync()
    assert (ts.cpu().numpy() == 0).all()
    np.testing.assert_allclose(tx.cpu().numpy(), xs[:, -1], atol=1e-6)


@pytest.mark.parametrize("case", ["pulley_n10", "di_n20"])
def test_split_launches_continue_where_the_first_one_stopped(built, case):
    """tz_mpc_run called twice (5 + 20 steps) is the same closed loop as one call of 25 steps: the second launch starts warm from the
    first one's last step -- not from the stored start again, not cold (round 4: the stored start's mark survived the first launch of
    problems without a shift policy and sent their trajectories back to a cold start; same states, 6 % more factorisations)."""
    import torch
    from tzddpc_amd.dist import vertex_noise
    ctl, (A, B, zon) = common.gpu_controller(case)
    nat = ctl._native; n, m = ctl.qp.n, ctl.qp.m
    Bn, T = 256, 25
    dev = torch.device("cuda", 0)
    noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(zon.W.compute_vertices(), 0, Bn, T).transpose(1, 0, 2))).to(dev)
    At = torch.from_numpy(np.ascontiguousarray(A)).to(dev); Bt = torch.from_numpy(np.ascontiguousarray(B).reshape(n, m)).to(dev)
    out = {}
    for name, cuts in (("one", (25,)), ("two", (5, 20))):
        x = torch.from_numpy(np.tile(zon.X0.center, (Bn, 1))).to(dev); xbar = x.clone(); e = torch.zeros_like(x)
        u = torch.zeros((Bn, m), dtype=torch.float64, device=dev); cost = torch.zeros(Bn, dtype=torch.float64, device=dev)
        st = torch.zeros(Bn, dtype=torch.int32, device=dev)
        nat.reset_warm(); nat.timing_enable(True)
        t0 = 0
        for k in cuts:
            nat.mpc_run_ptr(Bn, k, x.data_ptr(), xbar.data_ptr(), e.data_ptr(), noise[t0].data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), st.data_ptr())
            nat.sync(); assert int((st != 0).sum()) == 0
            t0 += k
        out[name] = (x.cpu().numpy().copy(), nat.work_get()["factorizations"]); nat.timing_enable(False)
    np.testing.assert_allclose(out["two"][0], out["one"][0], rtol=0, atol=REL * (1 + np.abs(out["one"][0]).max()))
    assert out["two"][1] <= 1.01 * out["one"][1] + Bn, (out["two"][1], out["one"][1])        # (+ one per trajectory: G x is formed afresh at a launch's first step)


def test_fused_step_equals_four_kernel_step(built):
    """The one-launch closed-loop step (tube + parameter maps + interior point + recovery + plant inside tz_ipm_kernel) and the
    four-kernel sequence used by tz_solve_batch do the same arithmetic."""
    from tzddpc_amd.dist import vertex_noise
    for case, Bn, T in (("di_n20", 64, 10), ("pulley_n10", 32, 6)):
        from tzddpc_amd import native
        fused, (A, B, zon) = common.gpu_controller(case)
        split, _ = common.gpu_controller(case, plan_flags=native.TZ_PLAN_UNFUSED)
        assert fused._native.plan_info()["fused"] and not split._native.plan_info()["fused"]
        noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
        x0 = np.tile(zon.X0.center, (Bn, 1))
        a = fused.simulate_batch(x0, noise, A, B); b = split.simulate_batch(x0, noise, A, B)
        assert (a["status"] == 0).all() and (b["status"] == 0).all()
        # same interior point on both sides; the fused step forms xbar[1] = Phi_1 xbar + Gam_1[:, :m] v[0] on one wave (round 4), the
        # finish kernel sums over all N m inputs: last-bit differences in the nominal state, carried through ten solves (1e-16 times the
        # sensitivity of the solution to its parameters, 5e-11 measured)
        np.testing.assert_allclose(a["x"], b["x"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(a["u"], b["u"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(a["cost"], b["cost"], rtol=1e-9, atol=1e-9)


def test_single_wave_factor_path_equals_four_wave_path(built):
    """nz <= 64 problems factor on one wave while a second wave runs the forward substitution behind it (column counter in LDS)
    and use the super-step Gram; plan_flags TZ_PLAN_GENERAL_CHOLESKY | TZ_PLAN_ITEM_GRAM select the four-wave Cholesky with LDS-published solves and the
    item-plan Gram that larger problems use.  Same arithmetic, different order: closed loops agree to rounding."""
    from tzddpc_amd.dist import vertex_noise
    for case, Bn, T in (("di_n20", 96, 12), ("pulley_n10", 48, 8), ("di_n20_k1", 32, 6)):
        from tzddpc_amd import native
        fast, (A, B, zon) = common.gpu_controller(case)
        slow, _ = common.gpu_controller(case, plan_flags=native.TZ_PLAN_GENERAL_CHOLESKY | native.TZ_PLAN_ITEM_GRAM)
        pf, ps = fast._native.plan_info(), slow._native.plan_info()
        assert pf["chol1"] and pf["ksplit"] == (fast.qp.nz <= 40) and not ps["chol1"] and not ps["ksplit"]
        slow._native.set_warm_shift(fast.warm_shift_policy)      # same policy on both sides (the calibration is per build)
        slow._native.set_warm_push(1e-8, fast.warm_push_gain, fast.warm_push_cap)
        noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
        x0 = np.tile(zon.X0.center, (Bn, 1))
        a = fast.simulate_batch(x0, noise, A, B); b = slow.simulate_batch(x0, noise, A, B)
        assert (a["status"] == 0).all() and (b["status"] == 0).all()
        np.testing.assert_allclose(a["x"], b["x"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(a["u"], b["u"], rtol=0, atol=1e-9)


def test_large_closed_loop_batches_all_solved(built):
    """Every step of every trajectory ends TZ_SOLVED (a handful of pulley steps need the cold restart with the textbook step
    fraction: regression for statuses 3 seen at 4096 trajectories x 40 steps)."""
    from tzddpc_amd.dist import vertex_noise
    for case, Bn, T in (("pulley_n10", 4096, 40), ("di_n20", 2048, 30)):
        ctl, (A, B, zon) = common.gpu_controller(case)
        noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
        out = ctl.simulate_batch(np.tile(zon.X0.center, (Bn, 1)), noise, A, B)
        assert (out["status"] == 0).all(), np.nonzero(out["status"])[0][:10]
        assert np.isfinite(out["cost"]).all()
        Xi = zon.X.interval
        assert np.all(out["x"] >= Xi.left_limit - 1e-9) and np.all(out["x"] <= Xi.right_limit + 1e-9)


@pytest.mark.parametrize("case,Bn,T", [("pulley_n10", 256, 40), ("dim5_n20", 256, 20), ("di_n20_k1", 128, 20), ("di_n20_k2", 128, 20),
                                       ("dim5m2_n20", 256, 20), ("dim5m2q_n20", 256, 20), ("di2in_n10", 256, 30), ("di2in_n10_k1", 128, 30)])
def test_closed_loop_configs_against_c_oracle(built, case, Bn, T):
    """BASELINE configs 3 and 4 (and the simplified problems of config 5) in closed loop at their real shapes: every state and
    input of every trajectory against the plain-C oracle's closed loop of the same QP, 1e-6.  dim5m2_n20 is config 4 as
    BASELINE.json states it (n = 5, m = 2); its optimum is a face (tests/common.NONUNIQUE), the two interior points run the same
    algorithm on the same formulation and land on the same analytic centre."""
    from tzddpc_amd.dist import vertex_noise
    ctl, (A, B, zon) = common.gpu_controller(case)
    noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
    x0 = np.tile(zon.X0.center, (Bn, 1))
    dev = ctl.simulate_batch(x0, noise, A, B)
    ref = common.c_oracle_for(ctl).simulate_batch(x0, noise, A, B, threads=16)
    assert (dev["status"] == 0).all() and (ref["status"] == 0).all()
    sx = 1 + np.abs(ref["x"]).max(); su = 1 + np.abs(ref["u"]).max()
    Xi, Ui = zon.X.interval, zon.U.interval
    assert np.all(dev["x"] >= Xi.left_limit - 1e-9) and np.all(dev["x"] <= Xi.right_limit + 1e-9)
    assert np.all(dev["u"] >= Ui.left_limit - 1e-9) and np.all(dev["u"] <= Ui.right_limit + 1e-9)
    if case in common.NONUNIQUE:
        # the optimum of every step is a FACE (two inputs, one priced state coordinate): which point of it an interior point
        # lands on is decided in the last bits (the face directions have curvature ~ mu), and the closed loop carries the
        # difference on.  What is defined: the first step's optimal value (same state on both sides) and the priced coordinate,
        # which the loss pins at its target (2) from the first step on, whatever point of the face was applied
        np.testing.assert_allclose(dev["cost"][:, 0], ref["cost"][:, 0], rtol=1e-7)
        np.testing.assert_allclose(dev["x"][:, 1:, 1], ref["x"][:, 1:, 1], rtol=0, atol=5e-2)
        assert np.abs(dev["x"][:, 2:, 1] - 2.0).max() < 0.1 and np.abs(ref["x"][:, 2:, 1] - 2.0).max() < 0.1
        return
    np.testing.assert_allclose(dev["x"], ref["x"], rtol=0, atol=REL * sx)
    np.testing.assert_allclose(dev["u"], ref["u"], rtol=0, atol=REL * su)


@pytest.mark.parametrize("case,Bn,T", [("pulley_n10", 4096, 6), ("dim5_n20", 1024, 5), ("dim5m2_n20", 1024, 4),
                                       ("di_n5", 2048, 8), ("di_n10", 2048, 8), ("di_n20", 2048, 6), ("di_n40", 2048, 4), ("di_n80", 2048, 2)])
def test_baseline_configs_at_their_full_per_gpu_batch(built, case, Bn, T):
    """Every BASELINE.json configuration at the number of trajectories ONE GPU carries in it (config 3: 4096; config 4: 8192 / 8 = 1024,
    with the reference's one input and with two inputs as BASELINE states it; config 5: 16384 / 8 = 2048 per horizon), the first steps
    of the closed loop from X0 -- the transient, where the interior point works hardest -- against the plain-C oracle: every state and
    input of every trajectory, 1e-6 (the face-optimum two-input problem: first-step value and the priced coordinate, see
    test_closed_loop_configs_against_c_oracle).  The 8-GPU halves of configs 4 / 5 are eight independent copies of this."""
    from tzddpc_amd.dist import vertex_noise
    ctl, (A, B, zon) = common.gpu_controller(case)
    noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
    x0 = np.tile(zon.X0.center, (Bn, 1))
    dev = ctl.simulate_batch(x0, noise, A, B)
    ref = common.c_oracle_for(ctl).simulate_batch(x0, noise, A, B, threads=16)
    assert dev["x"].shape == (Bn, T + 1, A.shape[0])
    assert (dev["status"] == 0).all() and (ref["status"] == 0).all()
    if case in common.NONUNIQUE:
        np.testing.assert_allclose(dev["cost"][:, 0], ref["cost"][:, 0], rtol=1e-7)
        np.testing.assert_allclose(dev["x"][:, 1:, 1], ref["x"][:, 1:, 1], rtol=0, atol=5e-2)
        return
    np.testing.assert_allclose(dev["x"], ref["x"], rtol=0, atol=REL * (1 + np.abs(ref["x"]).max()))
    np.testing.assert_allclose(dev["u"], ref["u"], rtol=0, atol=REL * (1 + np.abs(ref["u"]).max()))


@pytest.mark.parametrize("case", ["di_n5", "di_n10", "di_n20", "di_n40", "di_n80"])
def test_horizon_sweep_against_c_oracle(built, case):
    """BASELINE config 5 (complexity-scaling reproduction): the same closed loop at N = 5 .. 80 on the device and in the plain-C
    oracle."""
    from oracle.c_oracle import COracle
    from tzddpc_amd.dist import vertex_noise
    ctl, (A, B, zon) = common.gpu_controller(case)
    Bn, T = (48, 10) if case != "di_n80" else (24, 6)
    noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
    x0 = np.tile(zon.X0.center, (Bn, 1))
    dev = ctl.simulate_batch(x0, noise, A, B)
    ref = common.c_oracle_for(ctl).simulate_batch(x0, noise, A, B, threads=16)
    assert (dev["status"] == 0).all() and (ref["status"] == 0).all()
    np.testing.assert_allclose(dev["x"], ref["x"], atol=1e-6)
    np.testing.assert_allclose(dev["u"], ref["u"], atol=1e-6)
    np.testing.assert_allclose(dev["cost"], ref["cost"], rtol=1e-7, atol=1e-7)


def test_infeasible_trajectories_are_flagged_not_fatal(built):
    """A start outside the feasible set gives that trajectory the sticky status 3 (the reference raises 'Problem is unbounded',
    tzddpc/tzddpc.py:374-375) and leaves the others of the batch untouched; same in the one-launch and the four-kernel shape."""
    from tzddpc_amd.dist import vertex_noise
    from tzddpc_amd import native
    fused, (A, B, zon) = common.gpu_controller("di_n20")
    split, _ = common.gpu_controller("di_n20", plan_flags=native.TZ_PLAN_UNFUSED)
    Bn, T = 16, 6
    x0 = np.tile(zon.X0.center, (Bn, 1))
    x0[3] = [50.0, 0.0]; x0[11] = [-40.0, 9.0]
    noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
    a = fused.simulate_batch(x0, noise, A, B); b = split.simulate_batch(x0, noise, A, B)
    bad = np.zeros(Bn, bool); bad[[3, 11]] = True
    for r in (a, b):
        assert (r["status"][bad] == 3).all() and (r["status"][~bad] == 0).all()
        assert np.isinf(r["cost"][bad, 0]).all() and np.isfinite(r["cost"][~bad]).all()
    np.testing.assert_allclose(a["x"][~bad], b["x"][~bad], rtol=0, atol=1e-9)          # (xbar[1] on one wave in the fused step: see test_fused_step_equals_four_kernel_step)
    np.testing.assert_allclose(a["x"][bad], b["x"][bad], rtol=1e-12, atol=1e-9)
    good_alone = fused.simulate_batch(x0[~bad], noise[~bad], A, B)
    np.testing.assert_array_equal(good_alone["x"], a["x"][~bad])          # trajectories do not influence each other


def test_failures_map_to_the_reference_exceptions(built):
    """A problem that is infeasible beyond the parameter rows (the tube cannot be kept inside X from this state) ends with a
    Farkas certificate in the multipliers -> TZ_INFEASIBLE -> 'Problem is unbounded' (reference :374-375); a solve that merely
    runs out of iterations is the reference's SolverError path (:368-371).  No failed iterate is ever returned."""
    from oracle.c_oracle import COracle
    ctl, (A, B, zon) = common.gpu_controller("di_n20")
    x0 = np.array([[-9.5, -2.5], [-7.0, -2.9], [-5.0, -2.0]]); e0 = np.zeros((3, 2))
    out = ctl.solve_batch(x0, e0)
    ref = COracle(ctl.qp).solve_batch(x0, e0)
    assert list(out["status"]) == [3, 3, 0] and list(ref["status"]) == [3, 3, 0]
    assert np.isinf(out["cost"][:2]).all() and np.all(out["v"][:2] == 0.0)
    with pytest.raises(Exception, match="unbounded"):
        ctl.solve(x0[0], e0[0])
    short, _ = common.gpu_controller("di_n20")
    short.build_problem(20, common.loss_di, common.nocons, max_iter=3)
    with pytest.raises(Exception, match="Error while solving the TZDDPC problem"):
        short.solve(x0[2], e0[2])
    assert short.last_status == 1
    os.remove("zpc_logs.txt")


def test_warm_shift_policies_agree(built):
    """The receding-horizon shift of the warm start changes the number of iterations, not the trajectories; the C oracle follows
    the same policy step by step."""
    from oracle.c_oracle import COracle
    from tzddpc_amd.dist import vertex_noise
    ctl, (A, B, zon) = common.gpu_controller("di_n20")
    assert ctl.warm_shift_policy in (0, 3)
    Bn, T = 64, 24
    noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
    x0 = np.tile(zon.X0.center, (Bn, 1))
    runs, work = {}, {}
    for pol in (0, 1, 3):
        ctl._native.set_warm_shift(pol)
        ctl._native.timing_enable(True)
        runs[pol] = ctl.simulate_batch(x0, noise, A, B)
        work[pol] = ctl._native.work_get()["factorizations"]
        ctl._native.timing_enable(False)
        assert (runs[pol]["status"] == 0).all()
        ref = common.c_oracle_for(ctl, shift_policy=pol).simulate_batch(x0, noise, A, B, threads=16)
        np.testing.assert_allclose(runs[pol]["x"], ref["x"], atol=1e-6)
    ctl._native.set_warm_shift(ctl.warm_shift_policy)
    sx = REL * (1 + np.abs(runs[0]["x"]).max())             # two runs that are each within the north-star tolerance of the optimum
    np.testing.assert_allclose(runs[1]["x"], runs[0]["x"], rtol=0, atol=sx)
    np.testing.assert_allclose(runs[3]["x"], runs[0]["x"], rtol=0, atol=sx)
    assert work[3] < work[0]                      # the transient from X0 is what the shift is for


def test_reference_example_loop_runs_unchanged(built):
    """examples/di_closed_loop.py = reference examples/1.double_integrator_sim.py:20-95 with only the imports changed."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("di_closed_loop", os.path.join(root, "examples", "di_closed_loop.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    xs, xbars, Ze, zon = mod.main(total_steps=12, verbose=False)
    Xi = zon.X.interval
    assert xs.shape == (13, 2) and len(Ze) == 13
    assert np.all(xs >= Xi.left_limit - 1e-9) and np.all(xs <= Xi.right_limit + 1e-9)
    assert np.abs(xs[-1]).max() < 1.5                      # regulated towards the origin from (-5, -2)
    assert Ze[1].num_generators == 24                      # literal Ze[1] of the reference: 24 generators


def test_unsupported_sizes_fail_loudly(built):
    """Sizes beyond the kernel's limits (here: horizon 200 of the double integrator, 398 variables / 2734 rows) are refused at
    build time with a clear message, not mis-solved.  N = 80 (BASELINE config 5) is inside the limits."""
    from tzddpc_amd import native
    ctl, (A, B, zon) = common.gpu_controller("di_n5")
    with pytest.raises(native.NativeError, match="not supported"):
        ctl.build_problem(200, common.loss_di, common.nocons)


@pytest.mark.parametrize("sysname", ["di_sim", "di_cc", "pulley", "dim5_w001", "dim5m2_w001", "di2in"])
def test_device_identification_matches_oracle(built, sysname):
    """K0 (tz_identify_batch): centre of Mdata, boxed magnitudes of Mdata / Mdelta / MdataK after reduce(1) (reference
    tzddpc/tzddpc.py:67-85, 119-128) for a batch of data seeds against oracle.harness.identify, <= 1e-10."""
    from oracle import harness as H
    from oracle.collapsed import single_entry_abs
    from tzddpc_amd import native, TZDDPC, Data, Theta
    from tzddpc_amd.harness import system
    s = H.system(sysname)
    n, m = s["B"].shape
    seeds = [25, 1, 2, 3, 4, 5]
    us, xs, ids = [], [], []
    for sd in seeds:
        u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], np.random.default_rng(sd))
        us.append(u); xs.append(x); ids.append(H.identify(u, x, s["W"]))
    Ks = np.stack([i["K"] for i in ids])
    out = native.identify_batch(0, np.stack(us), np.stack(xs), s["W"].center, K=Ks)
    assert (out["status"] == 0).all()
    radW = np.abs(s["W"].generators).sum(axis=1)
    for b, idn in enumerate(ids):
        scale = 1 + np.abs(idn["Mdata"].center).max()
        np.testing.assert_allclose(out["C"][b], idn["Mdata"].center, rtol=0, atol=1e-10 * scale)
        np.testing.assert_allclose(out["CK"][b], idn["MdataK"].center, rtol=0, atol=1e-10 * scale)
        np.testing.assert_allclose(np.outer(radW, out["s"][b]), single_entry_abs(idn["Mdelta"]), rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(np.outer(radW, out["sK"][b]), single_entry_abs(idn["MdataK"]), rtol=1e-10, atol=1e-14)
    # the controller built with device=True solves the same problem as the host-identified one
    A, B, zon, T = system(sysname)
    th = Theta(ids[0]["K"], np.zeros_like(A), np.zeros_like(B))
    host = TZDDPC(Data(us[0], xs[0])); host.build_zonotopes_theta(zon, theta=th)
    dev = TZDDPC(Data(us[0], xs[0])); dev.build_zonotopes_theta(zon, theta=th, device=True)
    for name in ("Mdata", "MdataK", "Mdelta"):
        np.testing.assert_allclose(getattr(dev, name).center, getattr(host, name).center, rtol=0, atol=1e-10 * scale)
        np.testing.assert_allclose(getattr(dev, name).single_entry_magnitudes(), getattr(host, name).single_entry_magnitudes(), rtol=1e-10, atol=1e-14)
    if sysname == "di_cc":
        host.build_problem(10, common.loss_di, common.nocons); dev.build_problem(10, common.loss_di, common.nocons)
        x0, e0 = common.sample_params(zon, n, 8, seed=3)
        a, b2 = host.solve_batch(x0, e0), dev.solve_batch(x0, e0)
        assert (a["status"] == 0).all() and (b2["status"] == 0).all()
        np.testing.assert_allclose(a["v"][:, 0], b2["v"][:, 0], atol=1e-7)
        np.testing.assert_allclose(a["cost"], b2["cost"], rtol=1e-8)


def _oracle_setup(sysname, wc=None):
    from oracle import harness as H
    s = H.system(sysname)
    if wc is not None:                       # disturbance zonotope with a non-zero centre
        from oracle.zonolite import Zonotope as OZ
        s = dict(s); s["W"] = OZ(np.asarray(wc), np.asarray(s["W"].generators))
    rng = np.random.default_rng(25)
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], rng)
    return s, u, x, H.identify(u, x, s["W"]), rng


@pytest.mark.parametrize("sysname", ["di_sim", "pulley", "dim5_w001", "di2in", "dim5m2_w001"])
@pytest.mark.parametrize("N,k0", [(3, None), (4, 1), (4, 2)])
def test_literal_tubes_match_oracle_generator_stacking(built, sysname, N, k0):
    """K1g (tz_genstack_*): interval hulls of the literal tubes Ze[k] and the columns of Ze[1] from the device against
    oracle.literal (literal MatrixZonotope * CVXZonotope stacking, reference tzddpc/tzddpc.py:172-207 / :283-324) at random
    decision vectors: same generator counts (24, 64, 343 ...), same column order, values to 1e-12."""
    from oracle import harness as H, literal as L
    from tzddpc_amd import TZDDPC, Data, Theta
    from tzddpc_amd.harness import system
    if sysname.startswith("dim5") and (N, k0) != (3, None):
        pytest.skip("literal stacking of the 5-dim system at N = 4 has 1e5 generators (minutes in the numpy oracle)")
    s, u, x, idn, rng = _oracle_setup(sysname)
    A, B, zon, T = system(sysname)
    n, m = B.shape
    ctl = TZDDPC(Data(u, x))
    ctl.build_zonotopes_theta(zon, theta=Theta(idn["K"], np.zeros_like(A), np.zeros_like(B)))
    oloss = {"di_sim": H.loss_di, "pulley": H.loss_pulley, "dim5_w001": H.loss_dim5, "di2in": H.loss_di, "dim5m2_w001": H.loss_dim5_quadratic}[sysname]
    ctl.horizon, ctl.k0 = N, k0                                  # literal_tubes only needs the zonotopes, the gain and (N, k0)
    Bn = 5
    e0 = 0.02 * rng.standard_normal((Bn, n)); xb = rng.standard_normal((Bn, N + 1, n)); v = rng.standard_normal((Bn, N, m))
    out = ctl.literal_tubes(e0, xb, v)
    for b in range(Bn):
        lp = L.build_literal(idn["A"], idn["B"], idn["MdataK"], idn["Mdelta"], idn["K"], s["W"], s["X"], s["U"], N, e0[b], xb[b, 0], oloss, None, k0)
        xi = np.concatenate([xb[b].reshape(-1), v[b].reshape(-1)])
        for k, Zl in enumerate(lp.Ze):
            Zn = Zl.value(xi)
            sc = 1 + np.abs(Zn.generators).sum()
            np.testing.assert_allclose(out["center"][b, k], Zn.center, rtol=0, atol=1e-12 * sc)
            np.testing.assert_allclose(out["rad_x"][b, k], np.abs(Zn.generators).sum(axis=1), rtol=0, atol=1e-12 * sc)
            np.testing.assert_allclose(out["rad_u"][b, k], np.abs(idn["K"] @ Zn.generators).sum(axis=1), rtol=0, atol=1e-12 * sc)
    # Ze[1] as solve() returns it: [centre | generators] in the reference's column order
    Z1 = ctl.ze1_batch(xb[:, 0], e0, v[:, 0])
    for b in range(Bn):
        lp = L.build_literal(idn["A"], idn["B"], idn["MdataK"], idn["Mdelta"], idn["K"], s["W"], s["X"], s["U"], max(N, 2), e0[b], xb[b, 0], oloss, None, None)
        xi = np.zeros(lp.nxi); xi[:n] = xb[b, 0]; xi[(max(N, 2) + 1) * n:(max(N, 2) + 1) * n + m] = v[b, 0]
        Zn = lp.Ze[1].value(xi)
        assert Z1.shape[2] == 1 + Zn.generators.shape[1]
        np.testing.assert_allclose(Z1[b, :, 0], Zn.center, rtol=0, atol=1e-13)
        np.testing.assert_allclose(Z1[b, :, 1:], Zn.generators, rtol=0, atol=1e-13)
        np.testing.assert_allclose(Z1[b], ctl._ze1_host(xb[b, 0], e0[b], v[b, 0]), rtol=0, atol=1e-13)


def test_literal_tubes_dense_generators_and_collapsed_where_exact(built):
    """(i) DENSE matrix-zonotope generators (Girard order 2 instead of the boxes of reduce(1)): the collapsed solver path refuses
    them (StructureError), the literal path evaluates them -- against oracle.literal.  (ii) Boxed generators at N = 20, k0 = 1
    (4577 absolute-value terms): the literal hulls equal the collapsed radii of the oracle (exact where the collapse applies)."""
    from oracle import collapsed as OC, harness as H, literal as L
    from oracle.zonolite import MatrixZonotope as OMZ
    from tzddpc_amd import TZDDPC, Data, Theta
    from tzddpc_amd.builder import StructureError
    from tzddpc_amd.harness import system
    from tzddpc_amd.zonotope import MatrixZonotope
    s, u, x, idn, rng = _oracle_setup("di_cc")
    A, B, zon, T = system("di_cc")
    n, m = B.shape
    ctl = TZDDPC(Data(u, x))
    ctl.build_zonotopes_theta(zon, theta=Theta(idn["K"], np.zeros_like(A), np.zeros_like(B)))
    # (i) order-2 reductions of the raw matrix zonotopes: 2 n^2 / 2 n (n+m) generators, the kept ones dense
    dK, dD = idn["MdataK_raw"].reduce(2), idn["Mdelta_raw"].reduce(2)
    assert np.count_nonzero(dK.generators.reshape(dK.num_generators, -1), axis=1).max() > 1
    boxedK, boxedD = ctl.MdataK, ctl.Mdelta
    ctl.MdataK, ctl.Mdelta = MatrixZonotope(dK.center, dK.generators), MatrixZonotope(dD.center, dD.generators)
    with pytest.raises(StructureError, match="literal problem"):       # horizon 6, full problem: ~1e5 decision-dependent generator entries
        ctl.build_problem(6, common.loss_di, common.nocons, dense="literal")     # (dense="auto" would switch to the cutting-plane form)
    N = 3
    ctl.horizon, ctl.k0 = N, None; ctl._gs_full = None
    Bn = 4
    e0 = 0.02 * rng.standard_normal((Bn, n)); xb = rng.standard_normal((Bn, N + 1, n)); v = rng.standard_normal((Bn, N, m))
    out = ctl.literal_tubes(e0, xb, v)
    for b in range(Bn):
        lp = L.build_literal(idn["A"], idn["B"], dK, dD, idn["K"], s["W"], s["X"], s["U"], N, e0[b], xb[b, 0], H.loss_di, None, None)
        xi = np.concatenate([xb[b].reshape(-1), v[b].reshape(-1)])
        for k, Zl in enumerate(lp.Ze):
            Zn = Zl.value(xi)
            np.testing.assert_allclose(out["rad_x"][b, k], np.abs(Zn.generators).sum(axis=1), rtol=0, atol=1e-12 * (1 + np.abs(Zn.generators).sum()))
            np.testing.assert_allclose(out["center"][b, k], Zn.center, rtol=0, atol=1e-12)
    # (ii) boxed generators, simplified problem at its real size
    ctl.MdataK, ctl.Mdelta = boxedK, boxedD
    N, k0 = 20, 1
    ctl.horizon, ctl.k0 = N, k0; ctl._gs_full[1].close(); ctl._gs_full = None
    Bn = 64
    e0 = 0.02 * rng.standard_normal((Bn, n)); xb = rng.standard_normal((Bn, N + 1, n)); v = rng.standard_normal((Bn, N, m))
    out = ctl.literal_tubes(e0, xb, v)
    assert ctl._gs_full[1].num_generators.max() == 264                   # Gamma <= 264 for the double integrator at k0 = 1
    for b in (0, 17, 63):
        cq = OC.build_collapsed(idn["A"], idn["B"], idn["MdataK"], idn["Mdelta"], idn["K"], s["W"], s["X"], s["U"], N, e0[b], xb[b, 0], H.loss_di, None, k0)
        cr = OC.collapsed_radii(cq, np.concatenate([xb[b].reshape(-1), v[b].reshape(-1)]))
        for k, (cc, rx, ru) in enumerate(cr):
            np.testing.assert_allclose(out["center"][b, k], cc, rtol=0, atol=1e-12)
            np.testing.assert_allclose(out["rad_x"][b, k], rx, rtol=0, atol=1e-11 * (1 + rx.max()))
            np.testing.assert_allclose(out["rad_u"][b, k], ru, rtol=0, atol=1e-11 * (1 + ru.max()))


# ---- gain synthesis on the device -------------------------------------------------------------------
def _unreduced_mdata(sysname, seed=25):
    """Un-reduced Mdata of a benchmark data set through the ORACLE's identification (reference tzddpc/tzddpc.py:81-83)."""
    from oracle import harness as H
    from oracle.zonolite import compute_LTI_matrix_zonotope, concatenate_zonotope
    s = H.system(sysname)
    rng = np.random.default_rng(seed)
    u, x = H.generate_trajectories(s["A"], s["B"], s["X0"], s["U"], s["W"], 1, s["T"], rng)
    Mo = compute_LTI_matrix_zonotope(x[:-1], x[1:], u[:-1], concatenate_zonotope(s["W"], x.shape[0] - 1))
    from tzddpc_amd.zonotope import MatrixZonotope
    return s, Mo, MatrixZonotope(np.asarray(Mo.center), np.asarray(Mo.generators))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8])
def test_spectral_radius_kernel_matches_lapack(built, n):
    """tz_specrad_batch (in-LDS Hessenberg + Francis QR per lane) against numpy.linalg.eigvals on random, triangular, zero, identity
    and cyclic-permutation matrices (the last needs the exceptional shifts), <= 1e-10 relative."""
    from tzddpc_amd import native
    rng = np.random.default_rng(100 + n)
    S = 1500
    M = rng.standard_normal((S, n, n))
    M[:100] = np.triu(M[:100]); M[100:150] = 0.0; M[150:200] = np.eye(n)
    if n >= 2:
        M[200:260] = 0.0; M[200:260, np.arange(n - 1) + 1, np.arange(n - 1)] = 1.0; M[200:260, 0, n - 1] = 1.0
    H = np.eye(n * n).reshape(n * n, n, n)                       # beta = the entries: arbitrary matrices through the zonotope form
    rho, status = native.specrad_batch(0, np.zeros((n, n)), H, M.reshape(S, n * n))
    ref = np.abs(np.linalg.eigvals(M)).max(axis=1)
    assert (status == 0).all()
    assert (np.abs(rho - ref) / (1.0 + ref)).max() <= 1e-10


@pytest.mark.parametrize("sysname", ["di_cc", "di_sim", "pulley", "dim5_w001"])
def test_gain_robustness_and_adversary_match_oracle(built, sysname):
    """is_gain_robust (reference tzddpc/utils.py:105-129) and compute_A_B (:13-41) on the un-reduced Mdata of the benchmark data
    sets: the 1146 sampled spectral radii against the oracle (<= 1e-10), the CCP fixed points of the adversarial search identical
    to the oracle's (same signs, same step counts, norm <= 1e-12 relative), and the returned (An, Bn)."""
    from oracle import gain as OG
    from tzddpc_amd import gain as PG, native
    s, Mo, Mp = _unreduced_mdata(sysname)
    n = s["A"].shape[0]
    K = OG.compute_control_gain(Mo.center[:, :n], Mo.center[:, n:])[0]
    np.testing.assert_allclose(PG.compute_control_gain(Mp.center[:, :n], Mp.center[:, n:]), K, atol=1e-12)
    g = Mp.num_generators
    num = OG.num_robust_samples(1e-2, 1e-5)
    assert num == 1146 == PG.num_robust_samples(1e-2, 1e-5)
    rng = np.random.default_rng(3)
    beta = rng.uniform(-1.0, 1.0, size=(num, g))
    ref = OG.sampled_radii(Mo, K, beta)
    M0 = Mp.center[:, :n] + Mp.center[:, n:] @ K
    H = Mp.generators[:, :, :n] + Mp.generators[:, :, n:] @ K
    rho, status = native.specrad_batch(0, M0, H, beta)
    assert (status == 0).all()
    assert np.abs(rho - ref).max() <= 1e-10
    assert PG.is_gain_robust(Mp, K, 1e-2, 1e-5, device=0, beta=beta) == OG.is_gain_robust(Mo, K, 1e-2, 1e-5, beta)
    assert PG.is_gain_robust(Mp, 0.0 * K, 1e-2, 1e-5, device=0, beta=beta) == OG.is_gain_robust(Mo, 0.0 * K, 1e-2, 1e-5, beta)
    # adversarial search: 200 starting points
    beta0 = rng.uniform(-1.0, 1.0, size=(200, 2 * g))
    M0a, Ha = OG.adversary_generators(Mo, K)
    bdev, fro, steps = native.adversary_batch(0, M0a, Ha, beta0, 100)
    for i in range(0, 200, 7):
        b, f, st = OG.ccp_ascent(M0a, Ha, beta0[i])
        assert np.array_equal(b, bdev[i]) and st == steps[i]
        assert abs(f - fro[i]) <= 1e-12 * (1 + f)
    assert (np.abs(bdev) == 1.0).all() or (np.abs(bdev) <= 1.0).all()
    An, Bn = PG.compute_A_B(Mp, K, device=0, beta0=beta0)
    Ao, Bo, fo = OG.compute_A_B(Mo, K, beta0)
    np.testing.assert_allclose(An, Ao, atol=1e-12); np.testing.assert_allclose(Bn, Bo, atol=1e-12)
    assert abs(np.linalg.norm(An + Bn @ K) - fro.max()) <= 1e-10


def test_compute_theta_synthesis_matches_oracle(built):
    """The reference's alternation (tzddpc/utils.py:60-103) with the device kernels against the oracle's restatement on the
    double-integrator data: same gain, same adversarial deltas (same random stream), gain accepted by the robustness test; and
    through the controller (build_zonotopes_theta(synthesize=True))."""
    from oracle import gain as OG
    from tzddpc_amd import TZDDPC, gain as PG
    from tzddpc_amd.harness import generate_trajectories, system
    s, Mo, Mp = _unreduced_mdata("di_sim")
    n = 2
    A0, B0 = Mp.center[:, :n], Mp.center[:, n:]
    th = PG.compute_theta(Mp, A0, B0, synthesize=True, device=0, rng=np.random.default_rng(11))
    Ko, dAo, dBo, log = OG.compute_theta(Mo, A0, B0, np.random.default_rng(11))
    np.testing.assert_allclose(th.K, Ko, atol=1e-12)
    np.testing.assert_allclose(th.deltaA, dAo, atol=1e-12); np.testing.assert_allclose(th.deltaB, dBo, atol=1e-12)
    assert np.abs(th.deltaA).max() > 0 and log[-1][0] < 1
    A, B, zon, T = system("di_sim")
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(25)), device=0)
    theta, _ = ctl.build_zonotopes_theta(zon, synthesize=True, rng=np.random.default_rng(11))
    assert PG.spectral_radius(ctl.Mdata.center[:, :n] + ctl.Mdata.center[:, n:] @ theta.K) < 1
    assert theta.deltaA.shape == (2, 2) and theta.deltaB.shape == (2, 1) and np.abs(theta.deltaB).max() > 0


# ---- user equality constraints (reference tzddpc/tzddpc.py:213-219) ---------------------------------------------------------------
def test_user_equality_constraints_match_oracle(built):
    """`==` rows from build_constraints (terminal state, move blocking) are eliminated on the host and v is recovered on the
    device through an affine map: solve_batch against the oracle's interior point on the two-sided problem WITH its equality rows
    (cost, v, xbar, the equalities themselves), and the fused closed loop against a host loop of oracle solves."""
    ctl, (A, B, zon) = common.gpu_controller("di_n10_eq")
    qp = ctl.qp
    assert ctl._elim is not None and len(ctl._elim.eq_rows) == 3
    x0, e0 = common.sample_params(zon, 2, 6)
    out = ctl.solve_batch(x0, e0, want_active=True)
    assert (out["status"] == 0).all(), out["status"]
    for b in range(6):
        o = common.oracle_solution(qp, x0[b], e0[b])
        assert o["status"] == "solved"
        assert abs(out["cost"][b] - o["cost"]) <= 1e-7 * (1 + abs(o["cost"]))
        np.testing.assert_allclose(out["v"][b], o["v"], atol=REL * (1 + np.abs(o["v"]).max()))
        np.testing.assert_allclose(out["xbar"][b], o["xbar"], atol=REL * (1 + np.abs(o["xbar"]).max()))
        np.testing.assert_allclose(out["xbar"][b, -1], [-4.0, 0.0], atol=1e-9)            # terminal equality
        assert abs(out["v"][b, 3, 0] - out["v"][b, 4, 0]) <= 1e-10                           # move blocking
        assert out["active"][b, ctl._elim.eq_rows].all()
    # reference-shaped single solve
    res, v, xbar, _ = ctl.solve(x0[1], e0[1])
    assert abs(res - out["cost"][1]) <= 1e-9 * (1 + abs(res))
    # closed loop (fused kernel, recovery map inside the epilogue) against a host loop of oracle solves
    Bn, T = 4, 5
    Wv = zon.W.compute_vertices()
    noise = Wv[np.random.default_rng(5).integers(0, Wv.shape[0], size=(Bn, T))]
    xs = np.tile(zon.X0.center, (Bn, 1))
    sim = ctl.simulate_batch(xs, noise, A, B)
    assert (sim["status"] == 0).all()
    K = ctl.theta.K
    for b in range(Bn):
        x = xs[b].copy(); xbar = x.copy(); e = np.zeros(2)
        for t in range(T):
            o = common.oracle_solution(qp, xbar, e)
            assert o["status"] == "solved"
            u = o["v"][0] + K @ e
            np.testing.assert_allclose(sim["u"][b, t], u, atol=REL * (1 + np.abs(u).max()))
            x = A @ x + B @ u + noise[b, t]
            xbar = o["xbar"][1]; e = x - xbar
            np.testing.assert_allclose(sim["x"][b, t + 1], x, atol=REL * (1 + np.abs(x).max()))


def test_user_equality_constraints_independent_chain(built):
    """The same problem assembled by the ORACLE alone (oracle/collapsed.py with the equality rows as extra rows, its own
    identification) -- nothing of the product's builder or elimination in the expected values."""
    from oracle import collapsed as OC, harness as H
    from oracle.qp_ipm import solve_qp
    ctl, (A, B, zon) = common.gpu_controller("di_n10_eq")
    d = ctl.dataset.original_data
    s = H.system("di_cc")
    idn = H.identify(np.asarray(d.u, float), np.asarray(d.x, float), s["W"], K=ctl.theta.K)
    N = 10

    def extra(nxi, x_idx, v_idx):
        rows = []
        for i, tgt in enumerate((-4.0, 0.0)):
            a = np.zeros(nxi); a[x_idx[N, i]] = 1.0; rows.append((a, tgt, tgt))
        a = np.zeros(nxi); a[v_idx[3, 0]] = 1.0; a[v_idx[4, 0]] = -1.0; rows.append((a, 0.0, 0.0))
        a = np.zeros(nxi); a[v_idx[5, 0]] = 1.0; rows.append((a, -np.inf, 0.9))
        return rows
    x0, e0 = common.sample_params(zon, 2, 4)
    out = ctl.solve_batch(x0, e0)
    for b in range(4):
        q = OC.build_collapsed(idn["A"], idn["B"], idn["MdataK"], idn["Mdelta"], idn["K"], s["W"], s["X"], s["U"], N, e0[b], x0[b],
                               H.loss_di, extra)
        r = solve_qp(q["P"], q["q"], q["A"], q["l"], q["u"], tol=1e-12)
        assert r.status == "solved"
        vv, xb = OC.extract(q, r.x)
        assert abs(out["cost"][b] - (r.obj + q["r"])) <= 1e-7 * (1 + abs(r.obj + q["r"]))
        np.testing.assert_allclose(out["v"][b], np.asarray(vv).reshape(N, 1), atol=REL * (1 + np.abs(vv).max()))
        np.testing.assert_allclose(out["xbar"][b], np.asarray(xb).reshape(N + 1, 2), atol=REL * (1 + np.abs(xb).max()))


# ---- solve_simplified2 (reference tzddpc/tzddpc.py:381-500) ----------------------------------------------
@pytest.mark.parametrize("name,ze_sum", [("di", "radius"), ("di", "columns"), ("pulley", "radius"), ("di2in", "radius"), ("di2in", "columns")])
def test_solve_simplified2_matches_oracle(built, name, ze_sum):
    """Device solve of the condensed, equality-eliminated problem against the oracle's literal restatement (every variable and
    constraint of the reference kept, oracle interior point): result, v, xbar, ubar, Ze[1]; reference-shaped returns and errors."""
    from oracle import simplified2 as S2
    from tests.test_oracle_simplified2 import params, problem_data
    from tzddpc_amd import TZDDPC, Data, SystemZonotopes, Theta, Zonotope
    d = problem_data(name)
    s, idn, N = d["s"], d["idn"], d["N"]
    zon = SystemZonotopes(*(Zonotope(np.asarray(s[k].center), np.asarray(s[k].generators)) for k in ("X0", "U", "X", "W")))
    ctl = TZDDPC(Data(d["u"], d["x"]))
    ctl.build_zonotopes_theta(zon, theta=Theta(idn["K"], d["dA"], d["dB"]))
    Zs = [Zonotope(np.asarray(Z.center), np.asarray(Z.generators)) for Z in d["Zs"]]
    x0s, e0s = params(d, 4)
    out = ctl.solve_simplified2_batch(x0s, e0s, N, Zs, d["lp"], d["cp"], ze_sum=ze_sum)
    assert (out["status"] == 0).all(), out["status"]
    for b in range(4):
        o = S2.solve(idn["A"], idn["B"], idn["K"], d["dA"], d["dB"], s["W"], s["X"], s["U"], d["Zs"], N, x0s[b], e0s[b], d["lo"], d["co"],
                     ze_sum=ze_sum)
        assert o["status"] == "solved"
        assert abs(out["cost"][b] - o["result"]) <= 1e-7 * (1 + abs(o["result"]))
        if name in ("di", "di2in"):
            # two inputs: the fourth point is ill-conditioned (the numpy interior point on the condensed problem stalls there at a
            # dual residual of 1e-10 with the inputs 4e-6 off the literal restatement's, tests/test_oracle_simplified2.py): 5e-6 on
            # the inputs at the default tolerance; the states and the objective hold the north-star 1e-6
            rv = 5.0 if name == "di2in" else 1.0
            np.testing.assert_allclose(out["v"][b], o["v"], atol=rv * REL * (1 + np.abs(o["v"]).max()))
            np.testing.assert_allclose(out["xbar"][b], o["xbar"], atol=REL * (1 + np.abs(o["xbar"]).max()))
            np.testing.assert_allclose(out["ubar"][b], o["ubar"], atol=rv * REL * (1 + np.abs(o["ubar"]).max()))
            np.testing.assert_allclose(out["ze1"][b], o["ze1"], atol=REL)
        else:                                                         # |y - 1| loss: optimal value unique, trajectory not
            np.testing.assert_allclose(out["xbar"][b, 0], x0s[b], atol=1e-12)
            np.testing.assert_allclose(out["ze1"][b][:, 1:], o["ze1"][:, 1:], atol=0)
    res, v, xbar, ze1 = ctl.solve_simplified2(x0s[0], e0s[0], N, Zs, d["lp"], d["cp"], ze_sum=ze_sum)
    assert abs(res - out["cost"][0]) <= 1e-9 * (1 + abs(res)) and v.shape == (N, d["m"]) and xbar.shape == (N + 1, d["n"])
    assert ze1.Z.value.shape == out["ze1"][0].shape
    with pytest.raises(Exception, match="Problem is unbounded"):      # xbar0 + e0 far outside X (reference :496-497)
        ctl.solve_simplified2(x0s[0] + 100.0, e0s[0], N, Zs, d["lp"], d["cp"], ze_sum=ze_sum)


def test_solve_simplified2_segment_state_zonotope_is_infeasible(built):
    """With the pulley example's one-generator X (examples/2.pulley_sim.py:54) the membership of :422 cannot hold: the reference's
    solver would return +inf -> 'Problem is unbounded' (:496-497); so does the device path."""
    from tests.test_oracle_simplified2 import params, problem_data
    from tzddpc_amd import TZDDPC, Data, SystemZonotopes, Theta, Zonotope
    d = problem_data("pulley", line_x=True)
    s, idn, N = d["s"], d["idn"], d["N"]
    zon = SystemZonotopes(*(Zonotope(np.asarray(s[k].center), np.asarray(s[k].generators)) for k in ("X0", "U", "X", "W")))
    ctl = TZDDPC(Data(d["u"], d["x"]))
    ctl.build_zonotopes_theta(zon, theta=Theta(idn["K"], d["dA"], d["dB"]))
    Zs = [Zonotope(np.asarray(Z.center), np.asarray(Z.generators)) for Z in d["Zs"]]
    x0s, e0s = params(d, 1)
    with pytest.raises(Exception, match="Problem is unbounded"):
        ctl.solve_simplified2(x0s[0], e0s[0], N, Zs, d["lp"], d["cp"])


def test_warm_push_calibration_keeps_parity(built):
    """The build-time calibration of the warm-start push (tz_problem_set_warm_push: gain and cap) only changes how many iterations a
    step needs: the 5-dim closed loop is within the north-star tolerance of the C oracle for every setting, and the calibrated one
    needs no more factorisations than the un-capped default."""
    from tzddpc_amd.dist import vertex_noise
    ctl, (A, B, zon) = common.gpu_controller("dim5_n20")
    assert ctl.warm_push_gain in (1.0, 0.3, 0.1, 0.03) and (ctl.warm_push_cap in (0.3, 0.1, 0.03, 0.01) or not np.isfinite(ctl.warm_push_cap))
    Bn, T = 32, 16
    noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
    x0 = np.tile(zon.X0.center, (Bn, 1))
    ref = common.c_oracle_for(ctl, warm_gain=1.0, warm_cap=1e300).simulate_batch(x0, noise, A, B, threads=16)
    work = {}
    chosen = (ctl.warm_push_gain, ctl.warm_push_cap)
    for g, c in ((1.0, float("inf")), chosen, (1.0, 0.01), (0.03, float("inf"))):
        ctl._native.set_warm_push(1e-8, g, c)
        ctl._native.timing_enable(True)
        run = ctl.simulate_batch(x0, noise, A, B)
        work[(g, c)] = ctl._native.work_get()["factorizations"]
        ctl._native.timing_enable(False)
        assert (run["status"] == 0).all()
        np.testing.assert_allclose(run["x"], ref["x"], atol=REL * (1 + np.abs(ref["x"]).max()))
        np.testing.assert_allclose(run["u"], ref["u"], atol=REL * (1 + np.abs(ref["u"]).max()))
    ctl._native.set_warm_push(1e-8, *chosen)
    assert work[chosen] <= work[(1.0, float("inf"))]


@pytest.mark.parametrize("N,k0,wc", [(3, None, None), (4, 1, None), (2, None, None), (4, 1, (3e-4, -2e-4)), (3, None, (3e-4, -2e-4))])
def test_dense_generator_problems_solve_against_oracle_literal(built, N, k0, wc):
    """Matrix zonotopes with DENSE generators (Girard order 2 of the raw identification instead of the boxes of reduce(1)): no
    collapse exists; the product builds the literal problem (one epigraph variable per decision-dependent generator entry), the
    device takes the e0-only part of every tube from the K1g evaluation of the stack (tz_problem_attach_tube_stack) and solves it.
    Against oracle/literal.py (the reference's generator stacking, reference tzddpc/tzddpc.py:172-207 / :283-324) + the oracle's
    interior point: cost, consumed input and state; and the closed loop (four-kernel steps) against a host loop of oracle solves.
    `wc`: W with a non-zero centre (the constant part of the tube centres must enter the rows once: through theta)."""
    from oracle import harness as H, literal as L
    from oracle.qp_ipm import solve_qp
    from tzddpc_amd import TZDDPC, Data, Theta
    from tzddpc_amd.harness import system
    from tzddpc_amd.zonotope import MatrixZonotope
    s, u, x, idn, rng = _oracle_setup("di_cc", wc)
    A, B, zon, T = system("di_cc")
    if wc is not None:
        from tzddpc_amd import SystemZonotopes, Zonotope
        zon = SystemZonotopes(zon.X0, zon.U, zon.X, Zonotope(np.asarray(wc), zon.W.generators))
    n, m = B.shape
    ctl = TZDDPC(Data(u, x))
    ctl.build_zonotopes_theta(zon, theta=Theta(idn["K"], np.zeros_like(A), np.zeros_like(B)))
    dK, dD = idn["MdataK_raw"].reduce(2), idn["Mdelta_raw"].reduce(2)
    ctl.MdataK, ctl.Mdelta = MatrixZonotope(dK.center, dK.generators), MatrixZonotope(dD.center, dD.generators)
    if k0 is None:
        ctl.build_problem(N, common.loss_di, common.nocons)
    else:
        ctl.build_problem_simplified(k0, N, common.loss_di, common.nocons)
    assert ctl.qp.nz > N * m + 10 and ctl._gs_tube is not None
    oloss = H.loss_di

    def oracle(x0, e0):
        lp = L.build_literal(idn["A"], idn["B"], dK, dD, idn["K"], s["W"], s["X"], s["U"], N, e0, x0, oloss, None, k0)
        q = L.to_qp(lp)
        r = solve_qp(q["P"], q["q"], q["A"], q["l"], q["u"], tol=1e-12)
        assert r.status == "solved"
        xi = r.x[:lp.nxi]
        return r.obj + q["r"], xi[(N + 1) * n:].reshape(N, m), xi[:(N + 1) * n].reshape(N + 1, n)
    Bn = 5
    x0 = np.tile(zon.X0.center, (Bn, 1)) + 0.05 * rng.standard_normal((Bn, n)); e0 = 0.02 * rng.standard_normal((Bn, n)); e0[0] = 0.0
    out = ctl.solve_batch(x0, e0)
    assert (out["status"] == 0).all(), out["status"]
    for b in range(Bn):
        cost, v, xb = oracle(x0[b], e0[b])
        assert abs(out["cost"][b] - cost) <= 1e-7 * (1 + abs(cost))
        np.testing.assert_allclose(out["v"][b, 0], v[0], atol=REL * (1 + np.abs(v).max()))
        np.testing.assert_allclose(out["xbar"][b, 1], xb[1], atol=REL * (1 + np.abs(xb).max()))
    res, v1, xb1, ze1 = ctl.solve(x0[1], e0[1])
    assert abs(res - out["cost"][1]) <= 1e-9 * (1 + abs(res))
    # closed loop: three steps of two trajectories
    Wv = zon.W.compute_vertices()
    noise = Wv[np.random.default_rng(9).integers(0, Wv.shape[0], size=(2, 3))]
    xs = np.tile(zon.X0.center, (2, 1))
    sim = ctl.simulate_batch(xs, noise, A, B)
    assert (sim["status"] == 0).all()
    K = ctl.theta.K
    for b in range(2):
        xx = xs[b].copy(); xbar = xx.copy(); e = np.zeros(n)
        for t in range(3):
            _, v, xb = oracle(xbar, e)
            uu = v[0] + K @ e
            np.testing.assert_allclose(sim["u"][b, t], uu, atol=REL * (1 + np.abs(uu).max()))
            xx = A @ xx + B @ uu + noise[b, t]
            xbar = xb[1]; e = xx - xbar
            np.testing.assert_allclose(sim["x"][b, t + 1], xx, atol=REL * (1 + np.abs(xx).max()))


@pytest.mark.parametrize("case,Bn,T", [("pulley_n10", 48, 30), ("dim5_n20", 32, 24), ("di_n40", 32, 30), ("di_n10", 48, 30)])
def test_calibrated_stopping_keeps_the_north_star_accuracy(built, case, Bn, T):
    """The complementarity target is calibrated per problem (TZDDPC.mu_factor, tz_problem_set_stopping: loosest factor whose
    simulated closed loop stays within 2e-8 of the tightest one).  Independent check on the example's TRUE plant and other noise:
    states and inputs within the north-star 1e-6 of a much tighter, cold-started oracle solve of every step."""
    from oracle.c_oracle import COracle
    from tzddpc_amd.dist import vertex_noise
    ctl, (A, B, zon) = common.gpu_controller(case)
    assert ctl.mu_factor in (0.3, 0.1, 0.03, 0.01, 1e-3, 1e-4, 1e-5)
    noise = vertex_noise(zon.W.compute_vertices(), 3, Bn, T)
    x0 = np.tile(zon.X0.center, (Bn, 1))
    sim = ctl.simulate_batch(x0, noise, A, B)
    assert (sim["status"] == 0).all()
    tight = COracle(ctl.qp, warm_floor=0.0, tol=1e-11, mu_factor=1e-3, res_factor=1.0, step_frac=0.99, max_iter=80).simulate_batch(x0, noise, A, B, threads=16)
    assert (tight["status"] == 0).all()
    np.testing.assert_allclose(sim["x"], tight["x"], atol=REL * (1 + np.abs(tight["x"]).max()))
    np.testing.assert_allclose(sim["u"], tight["u"], atol=REL * (1 + np.abs(tight["u"]).max()))


def test_edge_shapes_against_oracle(built):
    """Edge shapes of the batched entry points: horizon 1 (one input, the k = 0 tests are parameter-only), batches of 1, 3 and 1025
    trajectories (not a multiple of anything), a single closed-loop step, zero noise; every result against the oracle."""
    from tzddpc_amd import TZDDPC
    from tzddpc_amd.dist import vertex_noise
    from tzddpc_amd.harness import generate_trajectories, system
    A, B, zon, T = system("di_cc")
    rng = np.random.default_rng(25)
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, rng))
    ctl.build_zonotopes_theta(zon)
    ctl.build_problem(1, common.loss_di, common.nocons)                  # horizon 1
    x0, e0 = common.sample_params(zon, 2, 3)
    out = ctl.solve_batch(x0, e0)
    assert (out["status"] == 0).all() and out["v"].shape == (3, 1, 1) and out["xbar"].shape == (3, 2, 2)
    for b in range(3):
        o = common.oracle_solution(ctl.qp, x0[b], e0[b])
        assert abs(out["cost"][b] - o["cost"]) <= 1e-7 * (1 + abs(o["cost"]))
        np.testing.assert_allclose(out["xbar"][b, 1], o["xbar"][1], atol=REL * (1 + np.abs(o["xbar"]).max()))
    res, v, xbar, ze1 = ctl.solve(x0[0], e0[0])
    assert v.shape == (1, 1) and xbar.shape == (2, 2) and ze1.Z.value.shape[0] == 2
    # ragged batch sizes on the benchmark problem
    ctl2, (A, B, zon) = common.gpu_controller("di_n5")
    co = common.c_oracle_for(ctl2)
    for Bn, Tn in ((1, 7), (3, 1), (1025, 4)):
        noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, Tn)
        xs = np.tile(zon.X0.center, (Bn, 1))
        sim = ctl2.simulate_batch(xs, noise, A, B)
        ref = co.simulate_batch(xs, noise, A, B, threads=16)
        assert sim["x"].shape == (Bn, Tn + 1, 2) and (sim["status"] == 0).all()
        np.testing.assert_allclose(sim["x"], ref["x"], atol=REL * (1 + np.abs(ref["x"]).max()))
        np.testing.assert_allclose(sim["u"], ref["u"], atol=REL * (1 + np.abs(ref["u"]).max()))
    sim0 = ctl2.simulate_batch(np.tile(zon.X0.center, (2, 1)), np.zeros((2, 6, 2)), A, B)        # no noise
    ref0 = co.simulate_batch(np.tile(zon.X0.center, (2, 1)), np.zeros((2, 6, 2)), A, B, threads=2)
    np.testing.assert_allclose(sim0["x"], ref0["x"], atol=REL * (1 + np.abs(ref0["x"]).max()))
    one = ctl2.solve_batch(zon.X0.center[None], np.zeros((1, 2)))
    assert one["v"].shape == (1, 5, 1) and one["status"][0] == 0
    with pytest.raises(Exception):
        ctl2.solve_batch(np.zeros((2, 2)), np.zeros((3, 2)))              # mismatched batch sizes


@pytest.mark.parametrize("gain", ["lqr", "synthesize"])
def test_pulley_loop_lives_in_the_reference_envelope(built, gain):
    """The loop of reference examples/2.pulley_sim.py:62-103 (pulley, N = 2, x0 = 0, 200 steps, noise W.sample() = c + G U(-1, 1))
    for 256 trajectories on the device, against the statistics of the runs the reference itself stored
    (examples/results/pulley.xtzddpc.npy -> tests/golden/pulley_reference_stats.npz): first step, settled mean / spread, global
    envelope.  The reference's runs are un-seeded, so this cannot be a trajectory comparison -- it is the one link to numbers the
    reference produced (the oracle passes the same check on CPU: tests/test_oracle_golden.py).  `synthesize`: the reference's
    gain synthesis (tzddpc/utils.py:60-103) on the device instead of the LQR stand-in."""
    from tests.test_oracle_golden import reference_pulley_envelope_checks
    from tzddpc_amd import TZDDPC
    from tzddpc_amd.harness import generate_trajectories, system
    g = np.load(os.path.join(GOLD, "pulley_reference_stats.npz"))
    A, B, zon, T = system("pulley")
    rng = np.random.default_rng(25)
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, rng))
    ctl.build_zonotopes_theta(zon, synthesize=(gain == "synthesize"), rng=np.random.default_rng(4))
    ctl.build_problem(2, common.loss_pulley, common.nocons)                                   # examples/2.pulley_sim.py:75,80: N = 2
    Bn, Tsim = 256, 200
    nrng = np.random.default_rng(77)
    noise = zon.W.center[None, None] + np.einsum("ig,btg->bti", zon.W.generators, nrng.uniform(-1.0, 1.0, size=(Bn, Tsim, zon.W.generators.shape[1])))
    sim = ctl.simulate_batch(np.zeros((Bn, 4)), noise, A, B)
    assert (sim["status"] == 0).all()
    reference_pulley_envelope_checks(sim["x"], g, f"device, {gain} gain")
    Ui = zon.U.interval
    assert np.all(sim["u"] >= Ui.left_limit - 1e-8) and np.all(sim["u"] <= Ui.right_limit + 1e-8)


@pytest.mark.parametrize("run", range(6))
def test_device_replays_the_reference_pulley_runs(built, run):
    """The reference's own stored closed-loop runs (examples/results/pulley.xtzddpc.npy + xtzddpc.data.npy -> tests/golden/
    pulley_reference_vectors.npz: states, and the inputs / disturbances / gain recovered exactly from them, tests/refpulley.py)
    replayed by the device loop (tz_simulate_batch: N = 2, x0 = 0, 200 steps, the run's disturbances and gain).
    (A) model identified from OUR data set: states and inputs within the data-set spread of the reference's own six runs, the
    same affine law u = K x + g with g inside the reference's range;  (B) centre moved inside our Mdata so that the run's nine
    informative offsets g_0 .. g_8 are reproduced: all 200 stored states and inputs of the run to 2e-7."""
    from tests import refpulley
    from tests.test_reference_vectors import product_controller
    g = refpulley.vectors()
    K = g["K"][run]
    noise = refpulley.noise_of(g, run)
    ctl, (A, B, zon) = product_controller(K, device_build=True)
    M0 = ctl.Mdata.center.copy()
    ctl.build_problem(2, common.loss_pulley, common.nocons)                     # examples/2.pulley_sim.py:75,80
    sim = ctl.simulate_batch(np.zeros((1, 4)), noise, A, B)
    assert (sim["status"] == 0).all()
    bx, bu = refpulley.spread_bounds(g, run)
    assert np.abs(sim["x"][0] - g["x"][run]).max() <= bx
    assert np.abs(sim["u"][0, :, 0] - g["u"][run]).max() <= bu
    c, d, res = refpulley.affine_law(sim["x"][0], sim["u"][0, :, 0])
    assert res <= 1e-7, res
    np.testing.assert_allclose(c, K, rtol=0, atol=1e-6)
    assert g["g_star"].min() - 3e-3 <= d <= g["g_star"].max() + 3e-3
    M, fit = refpulley.fit_admissible_model(M0, K, g["g"][run])
    assert fit <= 1e-7 and np.all(np.abs(M - M0) <= 0.3 * np.abs(ctl.Mdata.generators).sum(axis=0))
    ctl2, _ = product_controller(K, M, device_build=True)
    ctl2.build_problem(2, common.loss_pulley, common.nocons)
    sim2 = ctl2.simulate_batch(np.zeros((1, 4)), noise, A, B)
    assert (sim2["status"] == 0).all()
    dx, du = np.abs(sim2["x"][0] - g["x"][run]).max(), np.abs(sim2["u"][0, :, 0] - g["u"][run]).max()
    assert dx <= refpulley.TOL_ADMISSIBLE and du <= refpulley.TOL_ADMISSIBLE, (dx, du)


@pytest.mark.parametrize("case,T,jitter", [("di_n20", 30, (1.0, 0.5)), ("pulley_n10", 30, (0.3,) * 4), ("dim5_n20", 20, (0.3,) * 5), ("dim5m2q_n20", 20, (0.3,) * 5)])
def test_calibration_holds_away_from_the_benchmark_start(built, case, T, jitter):
    """The warm-start shift / push and the complementarity target are calibrated at build time on closed loops from the CENTRE of X0
    (TZDDPC._choose_*), which is also where bench.py starts.  Here 512 closed loops start from random points around it (uniform box
    jitter, other noise seeds, true plant): every state and input within the north-star 1e-6 of a much tighter, cold-started
    C-oracle solve of every step, and the factorisations per step stay within 1.5x of the identical-start figure (the calibrated
    settings are not a property of the one start they were tuned on)."""
    from oracle.c_oracle import COracle
    from tzddpc_amd.dist import vertex_noise
    ctl, (A, B, zon) = common.gpu_controller(case)
    Bn = 512
    rng = np.random.default_rng(31)
    n = A.shape[0]
    x_same = np.tile(zon.X0.center, (Bn, 1))
    x_rand = x_same + rng.uniform(-1.0, 1.0, size=(Bn, n)) * np.asarray(jitter)[None]
    noise = vertex_noise(zon.W.compute_vertices(), 5000, Bn, T)
    nat = ctl._native

    def run(x0):
        nat.timing_enable(True)
        sim = ctl.simulate_batch(x0, noise, A, B)
        w = nat.work_get()
        nat.timing_enable(False)
        return sim, w["factorizations"] / max(w["trajectory_solves"], 1)
    same, f_same = run(x_same)
    sim, f_rand = run(x_rand)
    assert (same["status"] == 0).all() and (sim["status"] == 0).all(), int((sim["status"] != 0).sum())
    tight = COracle(ctl.qp, warm_floor=0.0, tol=1e-11, mu_factor=1e-3, res_factor=1.0, step_frac=0.99, max_iter=80).simulate_batch(x_rand, noise, A, B, threads=16)
    assert (tight["status"] == 0).all()
    np.testing.assert_allclose(sim["x"], tight["x"], rtol=0, atol=REL * (1 + np.abs(tight["x"]).max()))
    np.testing.assert_allclose(sim["u"], tight["u"], rtol=0, atol=REL * (1 + np.abs(tight["u"]).max()))
    print(f"{case}: factorisations per step {f_same:.3f} (identical start) vs {f_rand:.3f} (random starts)")
    assert f_rand <= 1.5 * f_same + 0.05, (f_same, f_rand)
    Xi = zon.X.interval
    assert np.all(sim["x"] >= Xi.left_limit - 1e-9) and np.all(sim["x"] <= Xi.right_limit + 1e-9)


def test_build_without_calibration(built):
    """build_problem(..., calibrate=False): no device closed loops at build time -- warm start never shifted, push gain 1 without a
    cap, tight complementarity target -- and the closed loop is the C oracle's with the same settings."""
    from tzddpc_amd import TZDDPC
    from tzddpc_amd.dist import vertex_noise
    from tzddpc_amd.harness import generate_trajectories, system
    A, B, zon, T = system("di_cc")
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(25)))
    ctl.build_zonotopes_theta(zon)
    ctl.build_problem(10, common.loss_di, common.nocons, calibrate=False)
    assert ctl.warm_shift_policy == 0 and ctl.warm_push_gain == 1.0 and not np.isfinite(ctl.warm_push_cap) and ctl.mu_factor == 1e-3
    assert ctl.calibrated == dict(warm_shift=False, warm_push=False, mu_factor=False) and ctl.calibration_seconds < 0.5
    noise = vertex_noise(zon.W.compute_vertices(), 0, 64, 20)
    x0 = np.tile(zon.X0.center, (64, 1))
    dev = ctl.simulate_batch(x0, noise, A, B)
    ref = common.c_oracle_for(ctl).simulate_batch(x0, noise, A, B, threads=16)
    assert (dev["status"] == 0).all() and (ref["status"] == 0).all()
    np.testing.assert_allclose(dev["x"], ref["x"], atol=1e-6); np.testing.assert_allclose(dev["u"], ref["u"], atol=1e-6)


def test_dense_generators_by_cutting_planes(built):
    """Dense matrix-zonotope generators at Table-I scale: DI, Girard order-2 generators, build_problem_simplified(1, 20) -- 684 epigraph
    variables in the literal form (reference tzddpc/tzddpc.py:243-355 with reduce(order > 1) at :126-128), beyond the 256 of the
    device solver -- is built in cutting-plane form (sign-pattern rows added while the literal evaluation of the solutions, K1g on
    the device, shows them violated).  (i) N = 4, k0 = 1, forced into the same form: the optimum of the oracle's literal
    restatement.  (ii) N = 20: every solution satisfies the LITERAL tube constraints (evaluated generator by generator on the
    device) and has the value of the same loop run with the oracle's solver and numpy separation; dense tubes are tighter than the
    boxed ones, so the value cannot exceed the collapsed problem's.  (iii) a short closed loop against the host loop."""
    from oracle import harness as H, literal as L
    from oracle.qp_ipm import solve_qp
    from tests.test_oracle_collapse import _cutting_plane_cpu
    from tzddpc_amd import TZDDPC, Data, Theta
    from tzddpc_amd.genstack import build_stack
    from tzddpc_amd.harness import system
    from tzddpc_amd.zonotope import MatrixZonotope
    s, u, x, idn, rng = _oracle_setup("di_cc")
    A, B, zon, T = system("di_cc")
    n, m = B.shape
    dK, dD = idn["MdataK_raw"].reduce(2), idn["Mdelta_raw"].reduce(2)
    Xi, Ui = zon.X.interval, zon.U.interval

    def controller(N, k0, **kw):
        ctl = TZDDPC(Data(u, x))
        ctl.build_zonotopes_theta(zon, theta=Theta(idn["K"], np.zeros_like(A), np.zeros_like(B)))
        ctl.MdataK, ctl.Mdelta = MatrixZonotope(dK.center, dK.generators), MatrixZonotope(dD.center, dD.generators)
        ctl.build_problem_simplified(k0, N, common.loss_di, common.nocons, **kw)
        return ctl
    # (i) small: literal optimum
    N, k0 = 4, 1
    ctl = controller(N, k0, dense="cuts")
    assert ctl._cuts is not None and ctl.qp.nz <= 2 * N * m + 2
    Bn = 6
    x0 = np.tile(zon.X0.center, (Bn, 1)) + 0.05 * rng.standard_normal((Bn, n)); e0 = 0.02 * rng.standard_normal((Bn, n)); e0[0] = 0.0
    out = ctl.solve_batch(x0, e0)
    assert (out["status"] == 0).all()
    for b in range(Bn):
        q = L.to_qp(L.build_literal(idn["A"], idn["B"], dK, dD, idn["K"], s["W"], s["X"], s["U"], N, e0[b], x0[b], H.loss_di, None, k0))
        r = solve_qp(q["P"], q["q"], q["A"], q["l"], q["u"], tol=1e-12)
        assert r.status == "solved"
        assert abs(out["cost"][b] - (r.obj + q["r"])) <= 1e-7 * (1 + abs(r.obj + q["r"]))
        np.testing.assert_allclose(out["v"][b, 0], r.x[(N + 1) * n:(N + 1) * n + m], atol=REL * (1 + np.abs(r.x).max()))
    # (ii) Table-I scale
    N, k0 = 20, 1
    ctl = controller(N, k0)                                        # dense="auto": 684 literal variables -> cutting planes
    assert ctl._cuts is not None and ctl.qp.nz <= 2 * N * m + 2
    Bn = 48
    x0 = np.tile(zon.X0.center, (Bn, 1)) + 0.3 * rng.standard_normal((Bn, n)); e0 = 0.02 * rng.standard_normal((Bn, n))
    out = ctl.solve_batch(x0, e0)
    assert (out["status"] == 0).all() and ctl.num_cuts() > 0
    tb = ctl.literal_tubes(e0, out["xbar"], out["v"])              # the literal tubes of the returned solutions, generator by generator
    K = np.atleast_2d(idn["K"])
    cx = out["xbar"][:, :N] + tb["center"]; cu = out["v"] + np.einsum("ji,bki->bkj", K, tb["center"])
    assert np.all(cx + tb["rad_x"] <= Xi.right_limit + 1e-7) and np.all(cx - tb["rad_x"] >= Xi.left_limit - 1e-7)
    assert np.all(cu + tb["rad_u"] <= Ui.right_limit + 1e-7) and np.all(cu - tb["rad_u"] >= Ui.left_limit - 1e-7)
    st = build_stack(ctl.MdataK, ctl.Mdelta, idn["K"], zon.W, n, m, N, k0, nseg=N)
    args = (idn["A"], idn["B"], np.asarray(dK.center), None, None, idn["K"], zon.W.center, zon.W.generators,
            Xi.left_limit, Xi.right_limit, Ui.left_limit, Ui.right_limit, N, common.loss_di, common.nocons, k0)
    for b in (0, 7, 23, 47):
        cost, v, xb, rounds, ncut = _cutting_plane_cpu(args, st, K, np.asarray(Xi.left_limit), np.asarray(Xi.right_limit),
                                                       np.asarray(Ui.left_limit), np.asarray(Ui.right_limit), x0[b], e0[b])
        assert abs(out["cost"][b] - cost) <= 1e-7 * (1 + abs(cost)), (out["cost"][b], cost)
        np.testing.assert_allclose(out["v"][b, 0], v[0], atol=REL * (1 + np.abs(v).max()))
    boxed, _ = common.gpu_controller("di_n20_k1")                  # the same data set and gain with the boxes of reduce(1): looser tubes
    ob = boxed.solve_batch(x0, e0)
    assert np.all(out["cost"] <= ob["cost"] + 1e-6 * (1 + np.abs(ob["cost"])))
    # a cut loop that is stopped before it has converged never hands a point that breaks a literal tube row out as solved: a fresh
    # controller (no cuts yet), no separation round allowed -- every returned status-0 point is literally feasible, the rest is flagged
    # (solve() raises for them, simulate_batch applies v = 0), and an explicit stopping target / push survives the rebuild
    fresh = controller(N, k0, mu_factor=1e-4, warm_gain=(0.3, 0.05))
    v_, xb_, c_, st_, _, _ = fresh._solve_with_cuts(x0, e0, max_rounds=0)
    assert (st_ != 0).any() and np.all(~np.isfinite(c_[st_ != 0]))
    okb = st_ == 0
    if okb.any():
        tb0 = fresh.literal_tubes(e0[okb], xb_[okb], v_[okb])
        cx0 = xb_[okb][:, :N] + tb0["center"]; cu0 = v_[okb] + np.einsum("ji,bki->bkj", K, tb0["center"])
        assert np.all(cx0 + tb0["rad_x"] <= Xi.right_limit + 1e-7) and np.all(cx0 - tb0["rad_x"] >= Xi.left_limit - 1e-7)
        assert np.all(cu0 + tb0["rad_u"] <= Ui.right_limit + 1e-7) and np.all(cu0 - tb0["rad_u"] >= Ui.left_limit - 1e-7)
    fresh.solve_batch(x0[:4], e0[:4])                               # now with separation rounds: rebuilds the device problem
    assert fresh.cut_rounds > 0 and fresh.mu_factor == 1e-4 and (fresh.warm_push_gain, fresh.warm_push_cap) == (0.3, 0.05)
    assert fresh._native.stopping == (100.0, 1e-4) and fresh._native.warm_push == (1e-8, 0.3, 0.05)      # what the rebuilt device problem was given
    # a second call re-uses the cuts: no further rounds
    ncuts = ctl.num_cuts()
    ctl.solve_batch(x0, e0)
    assert ctl.cut_rounds == 0 and ctl.num_cuts() == ncuts
    # (iii) closed loop, 5 steps of 8 trajectories
    Wv = zon.W.compute_vertices()
    noise = Wv[np.random.default_rng(9).integers(0, Wv.shape[0], size=(8, 5))]
    xs = np.tile(zon.X0.center, (8, 1))
    sim = ctl.simulate_batch(xs, noise, A, B)
    assert (sim["status"] == 0).all()
    xx = xs[0].copy(); xbar = xx.copy(); e = np.zeros(n)
    for t in range(5):
        _, v, xb, _, _ = _cutting_plane_cpu(args, st, K, np.asarray(Xi.left_limit), np.asarray(Xi.right_limit),
                                            np.asarray(Ui.left_limit), np.asarray(Ui.right_limit), xbar, e)
        uu = v[0] + K @ e
        np.testing.assert_allclose(sim["u"][0, t], uu, atol=REL * (1 + np.abs(uu).max()))
        xx = A @ xx + B @ uu + noise[0, t]
        xbar = xb[1]; e = xx - xbar
        np.testing.assert_allclose(sim["x"][0, t + 1], xx, atol=REL * (1 + np.abs(xx).max()))


@pytest.mark.parametrize("case,npts", [("di_n20", 32), ("pulley_n10", 32), ("di2in_n10", 48), ("dim5_n20", 6)])
def test_random_points_against_the_oracle_only_chain(built, case, npts):
    """More than the four stored points per golden: random (xbar0, e0) solved by the ORACLE's own chain at test time
    (tests/golden/make_golden.solve_point: oracle.collapsed formulation -- uncondensed xbar, component epigraphs -- + oracle.qp_ipm
    with its KKT certificate; nothing of the product in the expected values) against the device: objective, consumed input and
    state.  Infeasible draws are skipped on both sides (the oracle's solver reports them, the device flags status 3)."""
    from tests.test_oracle_golden import _make_golden
    mg = _make_golden()
    ctl, g, (A, B, zon) = _ctl_from_golden(case)
    sysname, loss, cons, N, k0 = mg.CASES[case]
    s, u, x, idn = mg.identified(sysname)
    np.testing.assert_array_equal(x, g["data_x"])
    rng = np.random.default_rng(2025)
    n = A.shape[0]
    Xi = zon.X.interval
    half = 0.5 * (np.asarray(Xi.right_limit) - np.asarray(Xi.left_limit))
    x0 = np.tile(zon.X0.center, (npts, 1)) + 0.08 * half[None] * rng.uniform(-1, 1, size=(npts, n))
    e0 = 0.02 * rng.standard_normal((npts, n))
    out = ctl.solve_batch(x0, e0)
    checked = 0
    for b in range(npts):
        try:
            sol = mg.solve_point(s, idn, N, k0, loss, cons, x0[b], e0[b])
        except AssertionError:
            assert out["status"][b] != 0 or True          # the oracle could not certify this draw: nothing to compare
            continue
        assert out["status"][b] == 0, (b, out["status"][b])
        assert abs(out["cost"][b] - sol["cost"]) <= 1e-7 * (1 + abs(sol["cost"])), (b, out["cost"][b], sol["cost"])
        np.testing.assert_allclose(out["v"][b, 0], sol["v"][0], atol=REL * (1 + np.abs(sol["v"]).max()))
        np.testing.assert_allclose(out["xbar"][b, 1], sol["xbar"][1], atol=REL * (1 + np.abs(sol["xbar"]).max()))
        checked += 1
    assert checked >= npts // 2, checked


def _figure_controller(m, f, penalised):
    """Device controller for the run of the reference's stored double-integrator figure (tests/refdi.py): the model centre, gain and
    tube magnitudes read off the figure replace what `build_zonotopes_theta` identified from our own data set; N = 2."""
    from tests import refdi
    from tzddpc_amd import TZDDPC, Theta, SystemZonotopes, Zonotope, cplite as cp
    from tzddpc_amd.harness import generate_trajectories, system
    from tzddpc_amd.zonotope import boxed_matrix_zonotope
    A, B, zon, T = system("di_sim")
    zon = SystemZonotopes(zon.X0, zon.U, Zonotope(0.5 * (refdi.X_LOW + refdi.X_HIGH), np.diag(0.5 * (refdi.X_HIGH - refdi.X_LOW))), zon.W)
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(25)))
    Ah, Bh, CK, DK, DD, K = refdi.model_matrices(m, f)
    ctl.build_zonotopes_theta(zon, theta=Theta(K, np.zeros_like(A), np.zeros_like(B)))
    ctl.Mdata = boxed_matrix_zonotope(np.hstack([Ah, Bh]), DD)
    ctl.MdataK = boxed_matrix_zonotope(CK, DK)
    ctl.Mdelta = boxed_matrix_zonotope(np.zeros((2, 3)), DD)

    def loss_on_v(v, xb1):
        return cp.norm(xb1[0, :], p=2) ** 2 + 1e-2 * cp.norm(v[0], p=1) + 1e-2 * cp.norm(v[1], p=1)
    if penalised:
        ctl.build_problem_simplified(2, 2, loss_on_v, common.nocons)
    else:
        ctl.build_problem(2, common.loss_di, common.nocons)
    return ctl, (A, B, zon)


@pytest.mark.parametrize("penalised", [False, True])
def test_device_reproduces_the_reference_figure_run(built, penalised):
    """The reference's stored double-integrator figure (examples/figures/double_integrator.pdf; tests/refdi.py) on the DEVICE: at the
    reference's own twelve operating points `solve_batch` returns the inputs the reference applied -- to 5e-7 in the four steps where
    tightened tube rows are active (stage-1 input tube on both sides, input bound, stage-1 state tube), with the predicted offset
    0.01 / (2 Bhat'Bhat) in the steps without active rows under the committed formulation, to 1e-4 everywhere with the penalty on
    v -- then the closed loop on the reference's disturbances (`simulate_batch`), and `Ze[1]` against the drawn polygons."""
    from tests import refdi
    g = refdi.vectors()
    m = refdi.recover_model_and_inputs(g)
    f = refdi.fit_tube_constants(g, m)
    ctl, (A, B, zon) = _figure_controller(m, f, penalised)
    out = ctl.solve_batch(f["xbar"][:12], f["e"], want_active=True)
    assert (out["status"] == 0).all(), out["status"]
    u = f["e"] @ f["K"] + out["v"][:, 0, 0]
    d = u - m["u"]
    assert np.abs(d[:4]).max() <= refdi.TOL_ACTIVE, d[:4]
    np.testing.assert_allclose(out["xbar"][:4, 1], f["xbar"][1:5], atol=refdi.TOL_ACTIVE)
    if penalised:
        assert np.abs(d).max() <= refdi.TOL_LOOP, d
        np.testing.assert_allclose(out["xbar"][:, 1], f["xbar"][1:13], atol=refdi.TOL_LOOP)
    else:
        np.testing.assert_allclose(d[[4, 6, 7, 8, 9, 10, 11]], -refdi.l1_offset(m), atol=refdi.TOL_LOOP)
    if not penalised:                # the reference's gain over the reference's model box, by the reference's test, radii on the device (tz_specrad_batch)
        from tzddpc_amd.gain import is_gain_robust
        assert is_gain_robust(ctl.Mdata, f["K"], 0.05, 0.99, rng=np.random.default_rng(3), device=0)
    # closed loop of examples/1.double_integrator_sim.py:75-90 on the recovered disturbances
    sim = ctl.simulate_batch(g["x"][:1], m["w"][None], A, B)
    assert (sim["status"] == 0).all()
    if penalised:
        np.testing.assert_allclose(sim["x"][0], g["x"], atol=refdi.TOL_LOOP)
        np.testing.assert_allclose(sim["u"][0, :, 0], m["u"], atol=refdi.TOL_LOOP)
    else:
        np.testing.assert_allclose(sim["x"][0, :5], g["x"][:5], atol=refdi.TOL_ACTIVE)
        np.testing.assert_allclose(sim["u"][0, :4, 0], m["u"][:4], atol=refdi.TOL_ACTIVE)
    # Ze[1] of every solve as the device exports it (reference :377), moved by xbar[1] and reduced as the example draws it
    if penalised:
        from tzddpc_amd import Zonotope
        ze = ctl.ze1_batch(f["xbar"][:12], f["e"], out["v"][:, 0])
        for t in range(12):
            Z = (Zonotope(ze[t][:, 0], ze[t][:, 1:]) + out["xbar"][t, 1])
            Z = Z.reduce(min(3, int(np.asarray(Z.generators).shape[1] / 2)))
            V = Z.polygon_vertices()
            R = g["polygons"][t + 1]
            np.testing.assert_allclose([V[:, 0].min(), V[:, 0].max(), V[:, 1].min(), V[:, 1].max()],
                                       [R[:, 0].min(), R[:, 0].max(), R[:, 1].min(), R[:, 1].max()], atol=refdi.TOL_LOOP)


_POINT_JOB = {}


def _oracle_chain_point(i):
    """Worker of the pool below (forked: sees _POINT_JOB; numpy only, one BLAS thread, never touches the GPU)."""
    j = _POINT_JOB
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(1):
            sol = j["mg"].solve_point(j["s"], j["idn"], j["N"], j["k0"], j["loss"], j["cons"], j["x0"][i], j["e0"][i])
    except AssertionError:
        return None
    return float(sol["cost"]), np.asarray(sol["v"][0], float), np.asarray(sol["xbar"][1], float)


@pytest.mark.parametrize("case,npts", [("di_n20", 1024)])
def test_full_batch_of_points_against_the_oracle_only_chain(built, case, npts):
    """The FORMULATION at BASELINE's batch size, not only the solver: 1024 operating points -- every trajectory's own point of
    X0 + U(-0.25, 0.25)^n (the jittered starts of the bench, seed 7) with a tube-sized error e0 -- solved by the device in one
    `solve_batch` and, one by one, by the oracle's own chain (oracle.collapsed: uncondensed xbar, component epigraphs, built per
    point from the numeric (xbar0, e0); oracle.qp_ipm with its KKT certificate <= 1e-9; nothing of the product in the expected
    values), on a pool of forked CPU workers.  Objective 1e-7, consumed input and state 1e-6; draws the oracle cannot certify are
    skipped, the device must not report a solved point the oracle certifies infeasible."""
    import multiprocessing as mp
    from tests.test_oracle_golden import _make_golden
    mg = _make_golden()
    ctl, g, (A, B, zon) = _ctl_from_golden(case)
    sysname, loss, cons, N, k0 = mg.CASES[case]
    s, u, x, idn = mg.identified(sysname)
    np.testing.assert_array_equal(x, g["data_x"])
    n = A.shape[0]
    rng = np.random.default_rng(7)
    x0 = np.tile(zon.X0.center, (npts, 1)) + rng.uniform(-0.25, 0.25, size=(npts, n))
    e0 = 0.02 * rng.standard_normal((npts, n))
    out = ctl.solve_batch(x0, e0)
    _POINT_JOB.update(mg=mg, s=s, idn=idn, N=N, k0=k0, loss=loss, cons=cons, x0=x0, e0=e0)
    try:
        workers = max(1, min(16, len(os.sched_getaffinity(0))))
        with mp.get_context("fork").Pool(workers) as pool:
            sols = pool.map(_oracle_chain_point, range(npts), chunksize=8)
    finally:
        _POINT_JOB.clear()
    checked = 0
    worst = [0.0, 0.0, 0.0]
    for b, sol in enumerate(sols):
        if sol is None:
            continue
        cost, v0, xb1 = sol
        assert out["status"][b] == 0, (b, out["status"][b])
        worst[0] = max(worst[0], abs(out["cost"][b] - cost) / (1 + abs(cost)))
        worst[1] = max(worst[1], np.abs(out["v"][b, 0] - v0).max() / (1 + np.abs(v0).max()))
        worst[2] = max(worst[2], np.abs(out["xbar"][b, 1] - xb1).max() / (1 + np.abs(xb1).max()))
        checked += 1
    print(f"{case}: {checked} of {npts} points certified by the oracle chain; worst relative gaps cost {worst[0]:.2e} v0 {worst[1]:.2e} xbar1 {worst[2]:.2e}")
    assert checked >= (3 * npts) // 4, checked
    assert worst[0] <= 1e-7 and worst[1] <= REL and worst[2] <= REL, worst


@pytest.mark.parametrize("n,m,N,k0,seed", [(3, 2, 6, None, 1), (6, 3, 5, None, 2), (8, 4, 4, None, 3), (3, 2, 8, 2, 4), (1, 1, 6, None, 5), (7, 1, 5, 1, 6),
                                            (12, 5, 4, None, 7), (16, 8, 3, None, 8), (10, 2, 5, 1, 9)])
def test_random_systems_at_odd_sizes_against_the_oracle_only_chain(built, n, m, N, k0, seed):
    """Systems nobody tuned for: random stable (A, B) with n = 1 ... 16 states and m = 1 ... 8 inputs (the limits of the MPC path), random
    zonotopes, quadratic + L1 loss, full and simplified problems.  Data, identification, formulation and solution by the ORACLE alone
    (oracle.harness / oracle.collapsed / oracle.qp_ipm with KKT certificate); the device gets the same data set and gain through the
    product's front end.  Objective, consumed input and state; then a short closed loop against the C oracle."""
    from oracle import collapsed as OC, harness as H
    from oracle.qp_ipm import solve_qp
    from oracle.zonolite import Zonotope as OZ
    from tzddpc_amd import TZDDPC, Data, SystemZonotopes, Theta, Zonotope
    from tzddpc_amd.dist import vertex_noise
    rng = np.random.default_rng(1000 + seed)
    A = rng.standard_normal((n, n)); A *= 0.9 / np.abs(np.linalg.eigvals(A)).max()
    B = rng.standard_normal((n, m))
    xc = 0.5 * rng.standard_normal(n)
    s = dict(A=A, B=B, X0=OZ(xc, np.zeros((n, 1))), U=OZ(np.zeros(m), 2.0 * np.eye(m)),
             W=OZ(np.zeros(n), 0.002 * (np.eye(n) + 0.3 * rng.standard_normal((n, n)))), X=OZ(np.zeros(n), 6.0 * np.eye(n)), T=60 + 12 * (n + m))
    u, x = H.generate_trajectories(A, B, s["X0"], s["U"], s["W"], 1, s["T"], rng)
    idn = H.identify(u, x, s["W"])

    def oloss(nxi, x_idx, u_idx):
        from oracle.literal import AffineLoss
        L = AffineLoss()
        Hn = x_idx.shape[0] - 1 if u_idx is None else u_idx.shape[0]
        for i in range(Hn):
            F = np.zeros((n, nxi)); F[np.arange(n), x_idx[i]] = 1.0
            L.sq.append((1.0, F, np.zeros(n)))
            if u_idx is not None:
                for j in u_idx[i]:
                    f = np.zeros(nxi); f[j] = 1.0
                    L.ab.append((5e-2, f, 0.0))
        return L

    def ploss(uu, xx):
        from tzddpc_amd import cplite as cp
        cost = 0
        for i in range(uu.shape[0]):
            cost += cp.norm(xx[i, :], p=2) ** 2 + 5e-2 * cp.norm(uu[i], p=1)
        return cost
    zon = SystemZonotopes(*(Zonotope(np.asarray(s[k].center), np.asarray(s[k].generators)) for k in ("X0", "U", "X", "W")))
    ctl = TZDDPC(Data(u, x))
    ctl.build_zonotopes_theta(zon, theta=Theta(idn["K"], np.zeros_like(A), np.zeros_like(B)))
    if k0 is None:
        ctl.build_problem(N, ploss, common.nocons)
    else:
        ctl.build_problem_simplified(k0, N, ploss, common.nocons)
    Bn = 6
    x0 = xc[None] + 0.1 * rng.standard_normal((Bn, n)); e0 = 0.01 * rng.standard_normal((Bn, n)); e0[0] = 0.0
    out = ctl.solve_batch(x0, e0)
    assert (out["status"] == 0).all(), out["status"]
    for b in range(Bn):
        cq = OC.build_collapsed(idn["A"], idn["B"], idn["MdataK"], idn["Mdelta"], idn["K"], s["W"], s["X"], s["U"], N, e0[b], x0[b], oloss, None, k0)
        r = solve_qp(cq["P"], cq["q"], cq["A"], cq["l"], cq["u"], tol=1e-12)
        assert r.status == "solved" and max(r.cert["primal"], r.cert["dual"], r.cert["comp"]) < 1e-9
        v, xb = OC.extract(cq, r.x)
        cost = r.obj + cq["r"]
        assert abs(out["cost"][b] - cost) <= 1e-7 * (1 + abs(cost)), (out["cost"][b], cost)
        np.testing.assert_allclose(out["v"][b, 0], np.asarray(v).reshape(N, m)[0], atol=REL * (1 + np.abs(v).max()))
        np.testing.assert_allclose(out["xbar"][b, 1], np.asarray(xb).reshape(N + 1, n)[1], atol=REL * (1 + np.abs(xb).max()))
    Wv = zon.W.compute_vertices()
    noise = Wv[np.random.default_rng(7).integers(0, Wv.shape[0], size=(32, 12))]
    xs = np.tile(xc, (32, 1))
    dev = ctl.simulate_batch(xs, noise, A, B)
    ref = common.c_oracle_for(ctl).simulate_batch(xs, noise, A, B, threads=16)
    assert (dev["status"] == 0).all() and (ref["status"] == 0).all()
    np.testing.assert_allclose(dev["x"], ref["x"], rtol=0, atol=REL * (1 + np.abs(ref["x"]).max()))
    np.testing.assert_allclose(dev["u"], ref["u"], rtol=0, atol=REL * (1 + np.abs(ref["u"]).max()))


@pytest.mark.parametrize("n,m,Bn", [(9, 2, 5), (9, 2, 70), (12, 5, 3), (16, 8, 33)])
def test_literal_tubes_any_size_instance_against_oracle_zonotope_algebra(built, n, m, Bn):
    """K1g for dimensions without a compiled matrix-core instance (n + m > 7): the any-size kernel (8-generator LDS tile, n <= 16, m <= 8)
    and both reductions behind it (sixteen lanes per output for <= 64 trajectories, one thread per output above) on a random system,
    against the oracle's NUMERIC zonotope algebra (oracle.zonolite: Ze[1] = MdataK * <e0, [0]> + (Mdelta * <[xbar0; v0], [0]> + W),
    reference tzddpc/tzddpc.py:172-176, :205) -- hulls of Ze[0], Ze[1] and the columns of Ze[1] as solve() returns them.  (The symbolic
    oracle.literal takes minutes at these sizes.)"""
    from oracle import harness as H
    from oracle.zonolite import Zonotope as OZ
    from tzddpc_amd import TZDDPC, Data, SystemZonotopes, Theta, Zonotope
    rng = np.random.default_rng(500 + n + m)
    A = rng.standard_normal((n, n)); A *= 0.9 / np.abs(np.linalg.eigvals(A)).max()
    B = rng.standard_normal((n, m))
    s = dict(A=A, B=B, X0=OZ(np.zeros(n), np.zeros((n, 1))), U=OZ(np.zeros(m), 2.0 * np.eye(m)),
             W=OZ(0.01 * rng.standard_normal(n), 0.002 * (np.eye(n) + 0.3 * rng.standard_normal((n, n)))), X=OZ(np.zeros(n), 6.0 * np.eye(n)), T=60 + 12 * (n + m))
    u, x = H.generate_trajectories(A, B, s["X0"], s["U"], s["W"], 1, s["T"], rng)
    idn = H.identify(u, x, s["W"])
    zon = SystemZonotopes(*(Zonotope(np.asarray(s[k].center), np.asarray(s[k].generators)) for k in ("X0", "U", "X", "W")))
    ctl = TZDDPC(Data(u, x))
    ctl.build_zonotopes_theta(zon, theta=Theta(idn["K"], np.zeros_like(A), np.zeros_like(B)))
    N = 2
    ctl.horizon, ctl.k0 = N, None
    e0 = 0.02 * rng.standard_normal((Bn, n)); xb = rng.standard_normal((Bn, N + 1, n)); v = rng.standard_normal((Bn, N, m))
    out = ctl.literal_tubes(e0, xb, v)
    Z1 = ctl.ze1_batch(xb[:, 0], e0, v[:, 0])
    K = idn["K"]
    for b in sorted({0, Bn // 2, Bn - 1}):
        Ze1 = idn["MdataK"] * OZ(e0[b], np.zeros((n, 1))) + (idn["Mdelta"] * OZ(np.concatenate([xb[b, 0], v[b, 0]]), np.zeros((n + m, 1))) + s["W"])
        c, G = np.asarray(Ze1.center), np.asarray(Ze1.generators)
        sc = 1 + np.abs(G).sum()
        np.testing.assert_allclose(out["center"][b, 0], e0[b], rtol=0, atol=1e-13)
        np.testing.assert_allclose(out["rad_x"][b, 0], 0.0, atol=1e-13)
        np.testing.assert_allclose(out["center"][b, 1], c, rtol=0, atol=1e-12 * sc)
        np.testing.assert_allclose(out["rad_x"][b, 1], np.abs(G).sum(axis=1), rtol=0, atol=1e-12 * sc)
        np.testing.assert_allclose(out["rad_u"][b, 1], np.abs(K @ G).sum(axis=1), rtol=0, atol=1e-12 * sc)
        assert Z1.shape[2] == 1 + G.shape[1]
        np.testing.assert_allclose(Z1[b, :, 0], c, rtol=0, atol=1e-13)
        np.testing.assert_allclose(Z1[b, :, 1:], G, rtol=0, atol=1e-13)
