"""A/B of the stored start (tz_problem_store_start) on the driver window and on the 50-step run from X0: python tools/gpu_stored_start_ab.py case B"""
import sys, numpy as np
sys.path.insert(0, ".")
import torch
from tests import common
from tzddpc_amd.dist import vertex_noise
case = sys.argv[1]; Bn = int(sys.argv[2])
dev = torch.device("cuda", 0)
for ss in ("off", "auto"):
    ctl, (A, B, zon) = common.gpu_controller(case, stored_start=ss)
    nat = ctl._native; n, m = ctl.qp.n, ctl.qp.m
    noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(zon.W.compute_vertices(), 0, Bn, 50).transpose(1, 0, 2))).to(dev)
    At = torch.from_numpy(np.ascontiguousarray(A)).to(dev); Bt = torch.from_numpy(np.ascontiguousarray(B).reshape(n, m)).to(dev)
    u = torch.zeros((Bn, m), dtype=torch.float64, device=dev); cost = torch.zeros(Bn, dtype=torch.float64, device=dev); st = torch.zeros(Bn, dtype=torch.int32, device=dev)
    res = {}
    for name, (w, k) in {"window 5+20": (5, 20), "full 0+50": (0, 50)}.items():
        x = torch.from_numpy(np.tile(zon.X0.center, (Bn, 1))).to(dev); xbar = x.clone(); e = torch.zeros_like(x)
        nat.reset_warm()
        args = lambda t0: (x.data_ptr(), xbar.data_ptr(), e.data_ptr(), noise[t0].data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), st.data_ptr())
        if w: nat.mpc_run_ptr(Bn, w, *args(0)); nat.sync()
        best = 1e9
        nat.timing_enable(True); nat.mpc_run_ptr(Bn, k, *args(w)); nat.sync(); ms, cnt = nat.timing_get(1); wk = nat.work_get(); nat.timing_enable(False)
        res[name] = f"{ms / max(cnt, 1):.3f} ms, {wk['factorizations'] / max(wk['trajectory_solves'], 1):.3f} fact/step, slowest {wk['max_factorizations_one_trajectory']}, unsolved {int((st != 0).sum())}"
    print(case, "stored_start", ss, res)
