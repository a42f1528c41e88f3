"""Closed-loop throughput of any test case (not the bench line): python tools/gpu_throughput.py case B [steps] [warmup]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
from tests import common
from tzddpc_amd.dist import vertex_noise
case = sys.argv[1]; Bn = int(sys.argv[2]); K = int(sys.argv[3]) if len(sys.argv) > 3 else 30; W = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ctl, (A, B, zon) = common.gpu_controller(case)
nat = ctl._native; n, m = ctl.qp.n, ctl.qp.m
import os
if os.environ.get("TZ_SHIFT_POLICY"):
    ctl.warm_shift_policy = int(os.environ["TZ_SHIFT_POLICY"]); nat.set_warm_shift(ctl.warm_shift_policy)
if os.environ.get("TZ_QUIET"):
    nat.set_warm_quiet(int(os.environ["TZ_QUIET"]))
dev = torch.device("cuda", 0)
x = torch.from_numpy(np.tile(zon.X0.center, (Bn, 1))).to(dev); xbar = x.clone(); e = torch.zeros_like(x)
noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(zon.W.compute_vertices(), 0, Bn, K + W).transpose(1, 0, 2))).to(dev)
u = torch.zeros((Bn, m), dtype=torch.float64, device=dev); cost = torch.zeros(Bn, dtype=torch.float64, device=dev)
st = torch.zeros(Bn, dtype=torch.int32, device=dev)
At = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev); Bt = torch.from_numpy(np.ascontiguousarray(B, dtype=np.float64).reshape(n, m)).to(dev)
args = lambda t0: (x.data_ptr(), xbar.data_ptr(), e.data_ptr(), noise[t0].data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), st.data_ptr())
nat.mpc_run_ptr(Bn, W, *args(0)); nat.sync(); bad0 = int((st != 0).sum())
nat.timing_enable(True)
t0 = time.perf_counter(); nat.mpc_run_ptr(Bn, K, *args(W)); nat.sync(); dt = time.perf_counter() - t0
w = nat.work_get()
print(f"{case}: nz={ctl.qp.nz} rows={nat.mi} B={Bn}: {Bn * K / dt:,.0f} MPC steps/s ({dt / K * 1e3:.3f} ms per step of the batch), "
      f"{w['factorizations'] / max(w['trajectory_solves'], 1):.2f} factorisations per trajectory-step (warm-shift policy {ctl.warm_shift_policy}, push gain {ctl.warm_push_gain} cap {ctl.warm_push_cap}, mu {ctl.mu_factor}), unsolved {int((st != 0).sum())} (+{bad0} in warm-up), plan {nat.plan_info()}")
