"""Other identification data sets (seeds): every closed-loop step solved, device = C oracle, whatever the calibration picks."""
import sys, numpy as np
sys.path.insert(0, ".")
from tests import common
from tzddpc_amd.dist import vertex_noise
for case in ("di_n20", "di_n40", "pulley_n10", "dim5_n20", "di_n20_k1", "di_n10"):
    for seed in (1, 2, 3, 4):
        try:
            ctl, (A, B, zon) = common.gpu_controller(case, seed=seed)
        except Exception as ex:
            print(case, seed, "build failed:", str(ex)[:100]); continue
        Bn, T = 128, 30
        noise = vertex_noise(zon.W.compute_vertices(), 11, Bn, T)
        x0 = np.tile(zon.X0.center, (Bn, 1))
        sim = ctl.simulate_batch(x0, noise, A, B)
        ref = common.c_oracle_for(ctl).simulate_batch(x0[:32], noise[:32], A, B, threads=16)
        ok = ref["status"] == 0
        err = np.abs(sim["x"][:32][ok] - ref["x"][ok]).max() if ok.any() else float("nan")
        print(f"{case} seed {seed}: unsolved {int((sim['status'] != 0).sum())}/{Bn} (oracle {int((~ok).sum())}/32), max |x_dev - x_oracle| {err:.1e}, "
              f"shift {ctl.warm_shift_policy} push {ctl.warm_push_gain}/{ctl.warm_push_cap} mu {ctl.mu_factor}")
