"""Latency of the reference-style API: TZDDPC.solve() for one trajectory (host pointers, cold start)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from tests import common
for case in ("di_n5", "di_n20", "pulley_n10"):
    ctl, (A, B, zon) = common.gpu_controller(case)
    n = ctl.qp.n
    x0 = np.asarray(zon.X0.center, float); e0 = np.zeros(n)
    for _ in range(3): ctl.solve(x0, e0)
    t0 = time.perf_counter()
    for _ in range(50): out = ctl.solve(x0, e0)
    dt = (time.perf_counter() - t0) / 50
    xb, eb = np.tile(x0, (1024, 1)), np.zeros((1024, n))
    ctl.solve_batch(xb, eb)
    t0 = time.perf_counter()
    for _ in range(10): ctl.solve_batch(xb, eb)
    db = (time.perf_counter() - t0) / 10
    print(f"{case}: solve() {dt * 1e3:.3f} ms per call (one trajectory, cold start, host in/out); solve_batch(1024) {db * 1e3:.3f} ms")
