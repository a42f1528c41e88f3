"""Long closed loops: every step solved, device and C oracle stay together (drift of the carried G x, warm starts)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from tests import common
from oracle.c_oracle import COracle
from tzddpc_amd.dist import vertex_noise
for case, Bn, T in (("di_n20", 1024, 2000), ("pulley_n10", 1024, 1000), ("di_n40", 512, 600), ("dim5_n20", 512, 300), ("di_n5", 1024, 1000),
                    ("dim5m2q_n20", 256, 200), ("di2in_n10", 1024, 1000), ("dim5m2_n20", 128, 100)):      # round 3: two inputs
    ctl, (A, B, zon) = common.gpu_controller(case)
    noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
    x0 = np.tile(zon.X0.center, (Bn, 1))
    t0 = time.perf_counter(); dev = ctl.simulate_batch(x0, noise, A, B); dt = time.perf_counter() - t0
    ns = 48
    ref = common.c_oracle_for(ctl).simulate_batch(x0[:ns], noise[:ns], A, B, threads=16)
    err = np.abs(dev["x"][:ns] - ref["x"]).max(axis=(0, 2))
    print(f"{case}: {Bn} x {T} steps in {dt:.2f} s ({Bn * T / dt:,.0f} steps/s incl. host copies), unsolved {int((dev['status'] != 0).sum())}, "
          f"oracle unsolved {int((ref['status'] != 0).sum())}, max |x_dev - x_oracle| {err.max():.2e} (at step {err.argmax()}), last 100 steps {err[-100:].max():.2e}")
