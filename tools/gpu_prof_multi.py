"""Per-phase cycle sums of workgroup 0, averaged over the steps of ONE multi-step launch in steady state:
   TZ_PROF=1 python tools/gpu_prof_multi.py [case] [steps]"""
import os, sys, numpy as np
os.environ["TZ_PROF"] = "1"
sys.path.insert(0, ".")
import torch
from tests import common
from tzddpc_amd.dist import vertex_noise
names = ["formH", "chol", "solve", "gemvT", "gemvG", "elem", "total", "iters", "ch_upd", "ch_diag", "ch_panel", "ch_bar", "prologue", "epilogue", "gram_loop", "gram_red", "gram_bar", "gram_rmw", "tube", "warm", "top", "step",
         "rd_a(vin+bar)", "rd_b(Px|G'lam)", "rd_c(colsum+bar)", "epi_a(v)", "epi_b(cost,xbar1)", "maps_q", "warm_a(shift,Gx)", "warm_b(reduce)", "test(top->rd)", "t1(rp loop)", "t2(reduce)", "t3(mu,recip)", "x34", "x35", "x36", "x37"]
case = sys.argv[1] if len(sys.argv) > 1 else "di_n20"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctl, (A, B, zon) = common.gpu_controller(case)
nat = ctl._native
n, m = ctl.qp.n, ctl.qp.m
for Bn in (1, 1024):
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(np.tile(zon.X0.center, (Bn, 1))).to(dev); xbar = x.clone(); e = torch.zeros_like(x)
    noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(zon.W.compute_vertices(), 0, Bn, 20 + K).transpose(1, 0, 2))).to(dev)
    u = torch.zeros((Bn, m), dtype=torch.float64, device=dev); cost = torch.zeros(Bn, dtype=torch.float64, device=dev)
    st = torch.zeros(Bn, dtype=torch.int32, device=dev)
    At = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev); Bt = torch.from_numpy(np.ascontiguousarray(B, dtype=np.float64)).to(dev)
    args = lambda t0: (x.data_ptr(), xbar.data_ptr(), e.data_ptr(), noise[t0].data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), st.data_ptr())
    nat.mpc_run_ptr(Bn, 20, *args(0)); nat.sync()          # transient
    nat.mpc_run_ptr(Bn, K, *args(20)); nat.sync()
    pr = nat.debug_fetch(0, 6)
    print(f"{case} B={Bn} {K} steps in one launch, per step: " + " ".join(f"{nm}={pr[i] / K:.0f}" for i, nm in enumerate(names) if i < len(pr) and i != 7 and pr[i] > 0))
