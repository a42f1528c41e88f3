"""Generates tests/golden/*.npz with the ORACLE (numpy literal/collapsed algebra + interior point + KKT certificates).

The reference cannot be imported here (ModuleNotFoundError: cvxpy / pyzonotope / pydatadrivenreachability, see
oracle/__init__.py), so these are the build's own goldens: every stored solution carries its
KKT certificate, and the collapsed form they are solved in is proven equal to literal generator stacking in
tests/test_oracle_collapse.py.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    for case in ["di_n2", "di_n5", "di_n20", "di_n20_k1", "pulley_n10", "dim5_n20"]:
        ctl, qp, (A, B, zon) = common.identified_qp(case)
        x0, e0 = common.sample_params(zon, qp.n, 4, seed=11)
        rec = dict(data_u=ctl.dataset.original_data.u, data_x=ctl.dataset.original_data.x, K=ctl.theta.K, x0=x0, e0=e0,
                   v=[], xbar=[], cost=[], active=[], cert=[], slack=[], y=[])
        for b in range(4):
            ref = common.oracle_solution(qp, x0[b], e0[b], tol=1e-12)
            assert ref["status"] == "solved", (case, b, ref["status"])
            c = ref["cert"]
            assert max(c["primal"], c["dual"], c["comp"]) < 1e-9, (case, c)
            rec["v"].append(ref["v"]); rec["xbar"].append(ref["xbar"]); rec["cost"].append(ref["cost"])
            rec["active"].append(ref["active"]); rec["cert"].append([c["primal"], c["dual"], c["comp"]])
            rec["slack"].append(ref["slack"]); rec["y"].append(ref["y"])
        np.savez_compressed(os.path.join(OUT, f"{case}.npz"), **{k: np.array(v) for k, v in rec.items()})
        print(case, "cost", rec["cost"], "max cert", np.max(rec["cert"]))
    # G5: closed-loop double integrator (config-1 analogue: sim zonotopes, N = 2, 12 steps), fixed vertex noise
    from oracle.c_oracle import COracle
    from tzddpc_amd.dist import vertex_noise
    ctl, qp, (A, B, zon) = common.identified_qp("di_n2")
    noise = vertex_noise(zon.W.compute_vertices(), 0, 3, 12, seed_base=500)
    x0 = np.tile(zon.X0.center, (3, 1))
    sim = COracle(qp).simulate_batch(x0, noise, A, B)
    assert (sim["status"] == 0).all()
    # cross-check the first step of the C closed loop with the numpy interior point
    ref = common.oracle_solution(qp, x0[0], np.zeros(2))
    Kg = ctl.theta.K
    u0 = Kg @ np.zeros(2) + ref["v"][0]
    assert np.abs(sim["u"][0, 0] - u0).max() < 1e-7
    np.savez_compressed(os.path.join(OUT, "di_n2_closed_loop.npz"), data_u=ctl.dataset.original_data.u, data_x=ctl.dataset.original_data.x,
                        K=Kg, x0=x0, noise=noise, x=sim["x"], u=sim["u"], cost=sim["cost"])
    print("closed loop final states", sim["x"][:, -1])


if __name__ == "__main__":
    main()
