import sys, numpy as np
sys.path.insert(0, ".")
from tests import common
for case in sys.argv[1:]:
    ctl, (A, B, zon) = common.gpu_controller(case)
    nat = ctl._native
    Wv = zon.W.compute_vertices()
    for (Bn, T) in ((24, 48), (24, 25), (256, 25)):
        rng = np.random.default_rng(12345)
        noise = Wv[rng.integers(0, Wv.shape[0], size=(Bn, T))]
        x0 = np.tile(np.asarray(zon.X0.center, float), (Bn, 1))
        res = []
        for cand in (1.0, 0.3, 0.1, 0.03, 0.01, 0.003, 0.001, 0.0):
            nat.set_warm_push(1e-8, cand); nat.reset_warm() if hasattr(nat, "reset_warm") else None
            nat.timing_enable(True)
            _, _, _, status = nat.simulate_batch(x0, noise, A, B)
            w = nat.work_get()["factorizations"]
            nat.timing_enable(False)
            res.append((cand, round(w / (Bn * T), 3), int((status != 0).sum())))
        print(case, "shift", ctl.warm_shift_policy, "chosen", ctl.warm_push_gain, (Bn, T), res)
