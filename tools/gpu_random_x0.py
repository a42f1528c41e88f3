"""Robustness: closed loops from random initial states in X0 (uniform over the zonotope's box coordinates)."""
import sys, numpy as np
sys.path.insert(0, ".")
from tests import common
from tzddpc_amd.dist import vertex_noise
for case, Bn, T in (("di_n20", 4096, 40), ("pulley_n10", 4096, 40), ("di_n5", 4096, 40), ("di_n20_k1", 2048, 30)):
    ctl, (A, B, zon) = common.gpu_controller(case)
    rng = np.random.default_rng(7)
    Z = zon.X0
    x0 = Z.center[None, :] + rng.uniform(-1, 1, size=(Bn, Z.generators.shape[1])) @ Z.generators.T
    noise = vertex_noise(zon.W.compute_vertices(), 0, Bn, T)
    r = ctl.simulate_batch(x0, noise, A, B)
    Xi = zon.X.interval
    inside = np.all(r["x"] >= Xi.left_limit - 1e-9) and np.all(r["x"] <= Xi.right_limit + 1e-9)
    print(f"{case}: {Bn} random starts x {T} steps: unsolved trajectories {int((r['status'] != 0).sum())}, states inside X: {inside}, |x_T| mean {np.abs(r['x'][:, -1]).mean():.3f}")
