"""CPU-side checks of the product's host logic: cplite capture, builder structure, ABI surface, C oracle, sharding."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests import common
from tzddpc_amd import cplite as cp


def _vars(N=3, n=2, m=1):
    nsym = N * m + N * m + n
    Cv = np.zeros((N * m, nsym)); Cv[:, :N * m] = np.eye(N * m)
    x = cp.Affine(np.eye(nsym)[:N * n] if N * n <= nsym else np.zeros((N * n, nsym)), np.zeros(N * n), (N, n))
    return cp.Affine(Cv, np.zeros(N * m), (N, m)), x, nsym


def test_cplite_atoms_and_values():
    u, x, nsym = _vars()
    rng = np.random.default_rng(0)
    xi = rng.standard_normal(nsym)
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(x[i, :], p=2) ** 2 + 1e-2 * cp.norm(u[i], p=1)
    cost = cp.as_convex(cost, nsym)
    X = x.value_at(xi); U = u.value_at(xi)
    assert np.isclose(cost.value_at(xi), (X ** 2).sum() + 1e-2 * np.abs(U).sum())
    c2 = cp.as_convex(3.0 * cp.norm(x[1, 0] - 1, p=2) + cp.norm(x[0], "inf") + cp.sum_squares(2 * x[2] - 1), nsym)
    assert np.isclose(c2.value_at(xi), 3 * abs(X[1, 0] - 1) + np.abs(X[0]).max() + ((2 * X[2] - 1) ** 2).sum())
    Q = np.array([[2.0, 0.5], [0.5, 1.0]])
    assert np.isclose(cp.as_convex(cp.quad_form(x[0], Q), nsym).value_at(xi), X[0] @ Q @ X[0])
    con = x[:, 1] <= 10
    assert con.kind == "<=" and con.expr.shape == (3,)
    np.testing.assert_allclose((x @ np.array([[1.0], [2.0]])).value_at(xi)[:, 0], X @ [1.0, 2.0])
    np.testing.assert_allclose((np.array([[1.0, 2.0, 3.0]]) @ x).value_at(xi), np.array([[1.0, 2.0, 3.0]]) @ X)


def test_cplite_rejects_what_the_qp_path_cannot_express():
    u, x, nsym = _vars()
    assert cp.as_convex(cp.norm(x[0, :], p=2), nsym).soc       # second-order cone: recorded; the builder drops it on a free u ...
    with pytest.raises(cp.CpliteError):                        # ... and refuses it anywhere else
        common.identified_qp("di_n5", loss=lambda u, x: cp.norm(x[1, :], p=2))
    with pytest.raises(cp.CpliteError):                        # simplified problem: the loss sees v, not a free variable
        common.identified_qp("di2in_n10_k1", loss=lambda u, x: cp.norm(u[0], p=2))
    with pytest.raises(cp.CpliteError):                        # another term ties u down: the cone on u no longer vanishes
        common.identified_qp("di2in_n10", loss=lambda u, x: cp.norm(u[0], p=2) + cp.norm(u[0] - 1.0, p=2) ** 2)
    _, qp, _ = common.identified_qp("di2in_n10", loss=lambda u, x: common.loss_di(u, x) + cp.norm(u[0], p=2))
    _, qp0, _ = common.identified_qp("di2in_n10")
    assert qp.nz == qp0.nz and np.array_equal(qp.P, qp0.P) and np.array_equal(qp.q0, qp0.q0)      # vanished exactly
    with pytest.raises(cp.CpliteError):
        x[0, 0] * x[0, 1]
    with pytest.raises(cp.CpliteError):
        -1.0 * cp.norm(u[0], 1)
    with pytest.raises(cp.CpliteError):
        cp.norm(x[0], 3)


def test_builder_structure_and_reference_quirks():
    ctl, qp, (A, B, zon) = common.identified_qp("di_n5")
    N, n, m = 5, 2, 1
    assert qp.ntheta == 2 * n + N * (2 * n + m)
    assert list(qp.tube.power) == [0, 1, 2, 3, 4]               # full problem: term1[k-1] = M_K^k <e0>
    # k = 0 state rows are parameter-only (xbar0 + e0 in X, reference :191-195 at k = 0)
    assert len(qp.f0) >= 2 * n
    # free u (reference :160,222): its L1 term vanishes, so no epigraph variable for it exists
    assert not any(v.startswith("s[") for v in qp.var_names)
    # v[N-1] and xbar[N] are unconstrained by the loss (non-unique): P has no curvature on the last input
    assert np.allclose(qp.P[N * m - 1], 0.0)
    ctl2, qp2, _ = common.identified_qp("di_n20_k1")
    assert list(qp2.tube.power[:5]) == [0, 1, 2, 3, 3] and qp2.tube.pmax == 3     # simplified: powers saturate at k0 + 2
    assert sum(v.startswith("s[") for v in qp2.var_names) == 20                  # loss(v, xbar[1:]): |v_i| is real (:336)
    with pytest.raises(Exception):
        from tzddpc_amd.builder import build_parametric_qp
        build_parametric_qp(A, B, ctl.MdataK.center, ctl.MdataK.single_entry_magnitudes(), ctl.Mdelta.single_entry_magnitudes(),
                            ctl.theta.K, zon.W.center, zon.W.generators, [-1, -1], [1, 1], [-1], [1], 3, lambda u, x: None, None)


def test_structure_error_for_dense_generators():
    from tzddpc_amd import TZDDPC
    from tzddpc_amd.builder import StructureError
    from tzddpc_amd.harness import generate_trajectories, system
    A, B, zon, T = system("di_cc")
    ctl = TZDDPC.__new__(TZDDPC); ctl.device = 0; ctl._native = None; ctl.qp = None
    ctl.update_identification_data(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(3)))
    ctl.build_zonotopes_theta(zon)
    ctl.MdataK.generators[0][0, 1] = 0.3                        # a generator with two non-zeros: no collapse, the literal problem
    with pytest.raises(StructureError, match="literal problem"):   # ... which outgrows the device solver at this horizon
        ctl.build_problem(12, common.loss_di, common.nocons)


def test_header_symbols_are_exported(built):
    from tzddpc_amd import native
    hdr = open(os.path.join(common.__file__.rsplit("/tests/", 1)[0], "include", "tzddpc.h")).read()
    names = set(re.findall(r"^(?:int|const char\*)\s+(tz_\w+)\s*\(", hdr, flags=re.M))
    assert names == set(native.EXPORTED_SYMBOLS), names ^ set(native.EXPORTED_SYMBOLS)
    L = native.lib()
    for nm in names:
        assert hasattr(L, nm)
    assert L.tz_abi_version() == native.TZ_ABI_VERSION


def test_product_fails_loudly_without_gpu(built):
    """No CPU fallback: on a box without a HIP device problem creation raises (skipped where a GPU exists)."""
    from tzddpc_amd import native
    try:
        ndev = native.device_count()
    except native.NativeError:
        ndev = 0
    if ndev > 0:
        pytest.skip("GPU present")
    from tzddpc_amd import TZDDPC
    from tzddpc_amd.harness import generate_trajectories, system
    A, B, zon, T = system("di_cc")
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(3)))
    ctl.build_zonotopes_theta(zon)
    with pytest.raises(native.NativeError):
        ctl.build_problem(3, common.loss_di, common.nocons)
    with pytest.raises(Exception):
        ctl.solve(np.zeros(2), np.zeros(2))


@pytest.mark.parametrize("case", ["di_n5", "pulley_n10"])
def test_c_oracle_matches_numpy_oracle(built, case):
    from oracle.c_oracle import COracle
    ctl, qp, (A, B, zon) = common.identified_qp(case)
    x0, e0 = common.sample_params(zon, qp.n, 4)
    out = COracle(qp).solve_batch(x0, e0, threads=2, want_active=True)
    assert (out["status"] == 0).all()
    for b in range(4):
        ref = common.oracle_solution(qp, x0[b], e0[b])
        assert abs(out["cost"][b] - ref["cost"]) <= 1e-8 * (1 + abs(ref["cost"]))
        np.testing.assert_allclose(out["v"][b, 0], ref["v"][0], atol=1e-6)
        np.testing.assert_allclose(out["xbar"][b, 1], ref["xbar"][1], atol=1e-6)


def test_c_oracle_flags_parameter_infeasibility(built):
    from oracle.c_oracle import COracle
    ctl, qp, (A, B, zon) = common.identified_qp("di_n5")
    x0 = np.array([[50.0, 0.0]]); e0 = np.zeros((1, 2))          # xbar0 far outside X: reference raises 'Problem is unbounded'
    out = COracle(qp).solve_batch(x0, e0)
    assert out["status"][0] == 3 and np.isinf(out["cost"][0])


def test_shard_ranges_partition_the_batch():
    from tzddpc_amd.dist import shard_range, vertex_noise
    for total, ws in [(1024, 1), (1024, 8), (1000, 3), (7, 8)]:
        parts = [shard_range(total, ws, r) for r in range(ws)]
        assert parts[0][0] == 0 and parts[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1
    Wv = np.array([[1.0, 0.0], [-1.0, 0.0], [0.0, 1.0], [0.0, -1.0]])
    full = vertex_noise(Wv, 0, 6, 5)
    np.testing.assert_array_equal(full[2:5], vertex_noise(Wv, 2, 3, 5))     # noise does not depend on the sharding


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from tzddpc_amd.dist import shard_range, gather_results
rank = int(os.environ["RANK"]); ws = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=rank, world_size=ws)
total = 11
lo, hi = shard_range(total, ws, rank)
full = torch.arange(total * 3, dtype=torch.float64).reshape(total, 3)
got = gather_results(full[lo:hi].clone(), total)
assert torch.equal(got, full), (rank, got)
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_gather_results_world_size_2_gloo(tmp_path):
    """N > 1 path on CPU: two ranks each own a shard, one all-gather rebuilds the per-trajectory table in order."""
    root = common.__file__.rsplit("/tests/", 1)[0]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script), root, port], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


def test_horizon_shift_maps():
    """Shift maps derived from the variable / row names: v[k] <- v[k+1], tau[j] <- tau[j+1], the last step keeps its own."""
    from tzddpc_amd.builder import horizon_shift
    _, qp, _ = common.identified_qp("di_n20")
    sv, sr = horizon_shift(qp)
    names = qp.var_names
    assert names[sv[names.index("v[3,0]")]] == "v[4,0]" and names[sv[names.index("v[19,0]")]] == "v[19,0]"
    assert names[sv[names.index("tau[5]")]] == "tau[6]"
    rows = qp.row_names
    assert rows[sr[rows.index("Xub[4,1]")]] == "Xub[5,1]" and rows[sr[rows.index("tau[3]+-+")]] == "tau[4]+-+"
    assert len(set(sv.tolist())) == len(sv) - 2           # only the two last-step variables are the target of two sources


def test_equality_elimination_preserves_the_optimum():
    """builder.eliminate_equalities: the reduced inequality-only problem the device solves has the optimum of the two-sided problem
    with its `==` rows (oracle interior point on both), v comes back through x = x0 + Xn xbar0 + Z y, rows that lose all their
    coefficients turn into parameter tests, and problems without equality rows pass through untouched."""
    from oracle.qp_ipm import solve_qp
    from tzddpc_amd.builder import eliminate_equalities, theta_reference
    ctl, qp, (A, B, zon) = common.identified_qp("di_n10_eq")
    red, el = eliminate_equalities(qp)
    assert len(el.eq_rows) == 3 and red.nz == max(qp.nz - 3, qp.N * qp.m) and red.nc + len(el.eq_rows) <= qp.nc
    nv = qp.N * qp.m
    x0s, e0s = common.sample_params(zon, 2, 3)
    for b in range(3):
        o = common.oracle_solution(qp, x0s[b], e0s[b])
        th = theta_reference(qp, x0s[b], e0s[b])
        r = solve_qp(red.P, red.q0 + red.Qt @ th, red.A, red.l0 + red.Lt @ th, red.u0 + red.Ut @ th, tol=1e-12)
        assert o["status"] == "solved" == r.status
        x = el.x0 + el.Xn @ x0s[b] + el.Z @ r.x
        cost = r.obj + red.r0 + red.r1 @ x0s[b] + x0s[b] @ red.R2 @ x0s[b]
        assert abs(cost - o["cost"]) <= 1e-7 * (1 + abs(o["cost"]))
        np.testing.assert_allclose(x[:nv], o["v"].ravel(), atol=1e-7)
        np.testing.assert_allclose(qp.A[el.eq_rows] @ x, qp.u0[el.eq_rows] + qp.Ut[el.eq_rows] @ th, atol=1e-10)
    same, none = eliminate_equalities(common.identified_qp("di_n5")[1])
    assert none is None and same.nz == same.P.shape[0]


def test_bench_gpus_flag_spawns_the_ranks_dry_run():
    """`bench.py --gpus 2` outside torchrun starts two ranks itself (before anything touches a GPU) and relays rank 0's line; the
    CPU rehearsal (--dry-run --backend gloo) exercises that launch + shard + gather logic: n_gpus 2, 2 x B gathered rows, the
    world size read back from the process group, and the gathered table equal to the unsharded one."""
    import json
    root = common.__file__.rsplit("/tests/", 1)[0]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--batch", "33", "--dry-run",
                        "--backend", "gloo"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                                   # ONE JSON line, from rank 0
    ln = lines[0]
    assert ln["n_gpus"] == 2 and ln["world_size_read_back"] == 2 and ln["config"]["gathered_rows"] == 66
    assert ln["gather_matches_unsharded"] is True and len(ln["rank_window_ms"]) == 2
    # rank 0's build-time calibration is what every rank runs with; the line says whether the ranks had agreed on their own
    assert ln["calibration_identical_across_ranks"] is True and ln["calibration_adopted_from_rank0"] == [3.0, 0.1, 0.01, 1e-5]
    div = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--batch", "5", "--dry-run", "--backend", "gloo"],
                         env=dict(env, TZ_DRYRUN_DIVERGE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert div.returncode == 0, div.stderr[-2000:]
    dl = [json.loads(x) for x in div.stdout.splitlines() if x.startswith("{")][0]
    assert dl["calibration_identical_across_ranks"] is False and dl["calibration_adopted_from_rank0"][1] == 0.1      # rank 0's gain, not rank 1's 0.2
    # under a launcher whose world size disagrees with --gpus the bench refuses instead of reporting a wrong n_gpus
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run", "--backend", "gloo"],
                         env=dict(env, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


_SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from tests import common
from tzddpc_amd.dist import shard_range, gather_results, vertex_noise
from oracle.c_oracle import COracle
rank = int(os.environ["RANK"]); ws = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=rank, world_size=ws)
ctl, qp, (A, B, zon) = common.identified_qp("di_n5")
total, T = 13, 6                                   # uneven shards: 7 + 6
Wv = zon.W.compute_vertices()
lo, hi = shard_range(total, ws, rank)
co = COracle(qp)
def loop(first, count):
    out = co.simulate_batch(np.tile(zon.X0.center, (count, 1)), vertex_noise(Wv, first, count, T), A, B)
    assert (out["status"] == 0).all()
    return torch.from_numpy(np.concatenate([out["cost"].sum(axis=1)[:, None], out["x"][:, -1]], axis=1))
got = gather_results(loop(lo, hi - lo), total)
if rank == 0:
    assert torch.equal(got, loop(0, total)), "sharded closed loops differ from the unsharded batch"
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_sharded_closed_loops_equal_the_unsharded_batch_gloo(built, tmp_path):
    """The N > 1 data path end to end on CPU: two gloo ranks each run the closed loop of THEIR shard of trajectories (noise by
    global trajectory index; the plain-C oracle stands in for the device, which this container does not have), one all-gather
    of (cost, final state), and the table equals the unsharded run bit for bit -- trajectories are independent units."""
    root = common.__file__.rsplit("/tests/", 1)[0]
    script = tmp_path / "shard_worker.py"
    script.write_text(_SHARD_WORKER)
    port = str(31500 + os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script), root, port], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_bench_accepts_the_reference_harness_flags():
    """bench.py carries the flags of the reference's complexity harness (examples/1.double_integrator_computation_complexity.py:179-189:
    -m TZDDPC|STZDDPC|ZPC, -n evaluations, -ho horizon, -k0): parsing only (no GPU)."""
    root = common.__file__.rsplit("/tests/", 1)[0]
    sys.path.insert(0, root)
    import bench
    a = bench.parse_args(["-m", "STZDDPC", "-ho", "6", "-k0", "2", "-n", "3"])
    assert a.config == "di_n20" and a.horizon == 6 and a.k0_override == 2 and a.repeats == 3
    a = bench.parse_args(["-m", "TZDDPC"])
    assert a.horizon == 3 and a.k0_override is None and a.repeats == 11           # the harness's default horizon; full problem
    a = bench.parse_args([])
    assert a.config == "di_n20" and a.k0_override == "keep" and a.gpus == 1 and a.horizon == 0
    a = bench.parse_args(["--config", "genstack_dim5_k1", "--batch", "32"])
    assert a.config in bench.GENSTACK_CONFIGS
    with pytest.raises(SystemExit):
        bench.parse_args(["-m", "ZPC"])                                            # the comparator is another algorithm: refused, loudly
    with pytest.raises(SystemExit):
        bench.parse_args(["--backend", "gloo"])                                    # gloo only as the CPU rehearsal (--dry-run)


def test_zonotope_polygon_matches_the_vertex_enumeration():
    """``Z.reduce(order).polygon`` of the reference's plots (examples/1.double_integrator_sim.py:170,174): boundary of a 2-D zonotope,
    counter-clockwise, equal as a set to the hull of the 2^g enumerated points."""
    from scipy.spatial import ConvexHull
    from tzddpc_amd.zonotope import Zonotope
    rng = np.random.default_rng(3)
    for g in (1, 2, 3, 6, 9):
        Z = Zonotope(rng.standard_normal(2), rng.standard_normal((2, g)))
        V = Z.polygon_vertices()
        assert V.shape == (2 * g, 2)
        W = Z.compute_vertices()
        hull = W[ConvexHull(W).vertices] if g > 1 else W
        a = np.array(sorted(map(tuple, np.round(V, 10)))); b = np.array(sorted(map(tuple, np.round(hull, 10))))
        np.testing.assert_allclose(a, b, atol=1e-9)
        if g > 1:
            assert 0.5 * np.sum(V[:, 0] * np.roll(V[:, 1], -1) - np.roll(V[:, 0], -1) * V[:, 1]) > 0
    boxed = Zonotope([0.0, 0.0], np.array([[1.0, 0.0, 0.5], [0.0, 2.0, 0.5]])).reduce(1)
    assert boxed.polygon_vertices().shape == (4, 2)
    with pytest.raises(ValueError):
        Zonotope(np.zeros(3), np.eye(3)).polygon_vertices()
