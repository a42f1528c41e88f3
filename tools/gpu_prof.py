import os, sys, time, numpy as np
os.environ["TZ_PROF"] = "1"
sys.path.insert(0, ".")
from tests import common
names = ["formH", "chol", "solve", "gemvT", "gemvG", "elem/reduce", "total", "iters"]
for case in sys.argv[1:] or ["di_n20"]:
    ctl, (A, B, zon) = common.gpu_controller(case)
    n = ctl.qp.n
    for Bn in (1, 1024):
        x0, e0 = common.sample_params(zon, n, Bn, seed=3)
        ctl.solve_batch(x0, e0)
        t0 = time.perf_counter(); out = ctl.solve_batch(x0, e0); dt = time.perf_counter() - t0
        pr = ctl._native.debug_fetch(0, 6)
        it = pr[7]
        print(f"{case} B={Bn} wall {dt*1e3:.2f} ms iters(block0) {it:.0f} plan {ctl._native.plan_info()}")
        print("   cycles/iter: " + "  ".join(f"{nm} {pr[i]/max(it+1,1):.0f}" for i, nm in enumerate(names[:7])))
        if len(pr) > 8: print("   chol detail/iter: update %.0f diag %.0f panel %.0f barriers %.0f" % tuple(pr[8:12] / max(it, 1)))
