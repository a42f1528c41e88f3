"""On the GPU box: how the factorisation count of a simulated closed loop depends on the warm-start push gain, for the true plant
of the example and for the identified centre (what the build-time calibration sees): python tools/calib_probe.py case..."""
import sys, numpy as np
sys.path.insert(0, ".")
from tests import common
for case in sys.argv[1:]:
    ctl, (A, B, zon) = common.gpu_controller(case)
    nat = ctl._native
    n = ctl.dim_x
    Ac, Bc = ctl.Mdata.center[:, :n], ctl.Mdata.center[:, n:]
    Wv = zon.W.compute_vertices()
    AB = ctl.Mdata.sample(1, np.random.default_rng(4321))[0]
    AB2 = ctl.Mdata.center + 0.25 * (AB - ctl.Mdata.center)
    for plant, (Ap, Bp) in (("true", (A, B)), ("centre", (Ac, Bc)), ("member", (AB[:, :n], AB[:, n:])), ("quarter", (AB2[:, :n], AB2[:, n:]))):
        for (Bn, T, seed) in ((24, 48, 12345),):
            rng = np.random.default_rng(seed)
            noise = Wv[rng.integers(0, Wv.shape[0], size=(Bn, T))]
            x0 = np.tile(np.asarray(zon.X0.center, float), (Bn, 1))
            res = []
            for cand in (1.0, 0.3, 0.1, 0.03, 0.01, 0.003, 0.001):
                nat.set_warm_push(1e-8, cand)
                nat.timing_enable(True)
                _, _, _, status = nat.simulate_batch(x0, noise, Ap, Bp)
                w = nat.work_get()["factorizations"]
                nat.timing_enable(False)
                res.append((cand, round(w / (Bn * T), 3)))
            print(case, plant, (Bn, T, seed), res)
