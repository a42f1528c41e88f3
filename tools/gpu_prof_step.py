"""Per-phase cycle sums of workgroup 0 for closed-loop steps (fused and unfused): TZ_PROF=1 python tools/gpu_prof_step.py"""
import os, sys, numpy as np
os.environ["TZ_PROF"] = "1"
sys.path.insert(0, ".")
import torch
from tests import common
names = ["formH", "chol", "solve", "gemvT", "gemvG", "elem", "total", "iters", "ch_upd", "ch_diag", "ch_panel", "ch_bar", "prologue", "epilogue", "gram_loop", "gram_red", "gram_bar", "gram_rmw"]
case = sys.argv[1] if len(sys.argv) > 1 else "di_n20"
ctl, (A, B, zon) = common.gpu_controller(case)
nat = ctl._native
n, m = ctl.qp.n, ctl.qp.m
for Bn in (1, 1024):
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(np.tile(zon.X0.center, (Bn, 1))).to(dev); xbar = x.clone(); e = torch.zeros_like(x)
    Wv = zon.W.compute_vertices()
    from tzddpc_amd.dist import vertex_noise
    noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(Wv, 0, Bn, 24).transpose(1, 0, 2))).to(dev)
    u = torch.zeros((Bn, m), dtype=torch.float64, device=dev); cost = torch.zeros(Bn, dtype=torch.float64, device=dev)
    st = torch.zeros(Bn, dtype=torch.int32, device=dev)
    At = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev); Bt = torch.from_numpy(np.ascontiguousarray(B, dtype=np.float64)).to(dev)
    for t in range(24):
        nat.mpc_run_ptr(Bn, 1, x.data_ptr(), xbar.data_ptr(), e.data_ptr(), noise[t].data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), st.data_ptr())
        nat.sync()
        pr = nat.debug_fetch(0, 6)
        if t in (0, 6, 20, 23): print(f"{case} B={Bn} step {t}: " + " ".join(f"{nm}={pr[i]:.0f}" for i, nm in enumerate(names) if i < len(pr)))
