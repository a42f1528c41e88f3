"""Closed-loop accuracy of the device path against a tight cold-started solve of the C oracle (same inputs).
   python tests/diag/gpu_accuracy.py [case] ; solver knobs through TZ_* environment variables."""
import os, sys, numpy as np
sys.path.insert(0, ".")
from tests import common
from oracle.c_oracle import COracle
from tzddpc_amd.dist import vertex_noise
case = sys.argv[1] if len(sys.argv) > 1 else "di_n20"
nb, T = 256, 40
ctl, (A, B, zon) = common.gpu_controller(case)
qp = ctl.qp
x0 = np.tile(zon.X0.center, (nb, 1)); noise = vertex_noise(zon.W.compute_vertices(), 0, nb, T)
truth = COracle(qp, warm_floor=0.0, tol=1e-11, mu_factor=1e-3, step_frac=0.99, max_iter=80).simulate_batch(x0, noise, A, B, threads=16)
ctl._native.timing_enable(True)
r = ctl.simulate_batch(x0, noise, A, B)
w = ctl._native.work_get()
ex = np.abs(r["x"] - truth["x"]).max(axis=(0, 2)); eu = np.abs(r["u"] - truth["u"]).max(axis=(0, 2))
print(f"{case}: x err {ex.max():.2e} (step {ex.argmax()}) u err {eu.max():.2e}  truth ok {(truth['status'] == 0).all()} gpu ok {(r['status'] == 0).all()} "
      f"factorisations/step {w['factorizations'] / max(w['trajectory_solves'], 1):.3f}  scale {np.abs(truth['x']).max():.2f}")
