"""Diagnostic: the degenerate two-input 5-dim golden points on the device under solver-option variants (status / iterations / cost)."""
import os, sys, numpy as np
sys.path.insert(0, ".")
from tests import common
from tzddpc_amd import TZDDPC, Data, Theta
from tzddpc_amd.harness import system
case = sys.argv[1] if len(sys.argv) > 1 else "dim5m2_n20"
g = np.load(f"tests/golden/{case}.npz")
sysname, loss, cons, N, k0 = common.CASES[case]
A, B, zon, T = system(sysname)
for kw in (dict(), dict(tol=1e-9), dict(tol=1e-8), dict(step_frac=0.99), dict(reg=1e-9), dict(reg=1e-7), dict(max_iter=80)):
    ctl = TZDDPC(Data(g["data_u"], g["data_x"]))
    ctl.build_zonotopes_theta(zon, theta=Theta(g["K"], np.zeros_like(A), np.zeros_like(B)))
    ctl.build_problem(N, loss, cons, calibrate=False, **kw)
    for mu in (1e-3, 0.1):
        ctl._native.set_stopping(100.0, mu)
        out = ctl.solve_batch(g["x0"], g["e0"])
        lam = ctl._native.debug_fetch(0, 5); s = ctl._native.debug_fetch(0, 4)
        print(kw, "mu_factor", mu, "status", out["status"], "iters", out["iters"], "cost rel err", np.abs(out["cost"] - g["cost"]) / (1 + np.abs(g["cost"])),
              "traj0: mean s*lam %.2e max lam %.2e min s %.2e" % (float((s * lam).mean()), float(lam.max()), float(s.min())), flush=True)
