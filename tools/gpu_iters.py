"""Iteration histogram per closed-loop step of the bench workload: python tools/gpu_iters.py [steps]"""
import os, sys, numpy as np
sys.path.insert(0, ".")
import torch
import bench
from tzddpc_amd.dist import vertex_noise
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctl, A, B, zon = bench.build_controller(0, 20)
nat = ctl._native; n, m = ctl.qp.n, ctl.qp.m; Bn = 1024
dev = torch.device("cuda", 0)
x = torch.from_numpy(np.tile(zon.X0.center, (Bn, 1))).to(dev); xbar = x.clone(); e = torch.zeros_like(x)
noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(zon.W.compute_vertices(), 0, Bn, steps).transpose(1, 0, 2))).to(dev)
u = torch.zeros((Bn, m), dtype=torch.float64, device=dev); cost = torch.zeros(Bn, dtype=torch.float64, device=dev)
st = torch.zeros(Bn, dtype=torch.int32, device=dev)
At = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev); Bt = torch.from_numpy(np.ascontiguousarray(B, dtype=np.float64)).to(dev)
tot = 0; win = 0; per = np.zeros(Bn, dtype=np.int64)
for t in range(steps):
    nat.mpc_run_ptr(Bn, 1, x.data_ptr(), xbar.data_ptr(), e.data_ptr(), noise[t].data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), st.data_ptr())
    nat.sync()
    it = nat.last_iterations(Bn)
    tot += it.sum()
    if t >= 5: win += it.sum(); per += it
    print(f"step {t:3d}: mean {it.mean():5.2f} max {it.max():3d} hist {np.bincount(it, minlength=1).tolist()}  |x| mean {float(x.abs().mean()):.3f}")
print("mean iterations", tot / (steps * Bn), "steps>=5:", win / (max(steps - 5, 1) * Bn), "bad", int((st != 0).sum()))
# load balance of the one-launch closed loop: the launch ends with its slowest trajectory (all 1024 workgroups are resident at once)
print(f"iterations per trajectory over steps >= 5: mean {per.mean():.1f} min {per.min()} max {per.max()} (max/mean {per.max() / per.mean():.2f}); "
      f"by CU (trajectories b, b+256, b+512, b+768 share one): mean of the CU maxima {per.reshape(4, 256).max(axis=0).mean():.1f}, CU sums max/mean {per.reshape(4, 256).sum(axis=0).max() / per.reshape(4, 256).sum(axis=0).mean():.2f}")
