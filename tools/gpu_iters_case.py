import sys, numpy as np
sys.path.insert(0, ".")
import torch
from tests import common
from tzddpc_amd.dist import vertex_noise
case = sys.argv[1]
ctl, (A, B, zon) = common.gpu_controller(case)
nat = ctl._native; n, m = ctl.qp.n, ctl.qp.m; Bn = 256; steps = 40
dev = torch.device("cuda", 0)
x = torch.from_numpy(np.tile(zon.X0.center, (Bn, 1))).to(dev); xbar = x.clone(); e = torch.zeros_like(x)
noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(zon.W.compute_vertices(), 0, Bn, steps).transpose(1, 0, 2))).to(dev)
u = torch.zeros((Bn, m), dtype=torch.float64, device=dev); cost = torch.zeros(Bn, dtype=torch.float64, device=dev)
st = torch.zeros(Bn, dtype=torch.int32, device=dev)
At = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev); Bt = torch.from_numpy(np.ascontiguousarray(B, dtype=np.float64).reshape(n, m)).to(dev)
for pol in (0, 3, 1):
    nat.set_warm_shift(pol)
    x.copy_(torch.from_numpy(np.tile(zon.X0.center, (Bn, 1)))); xbar.copy_(x); e.zero_()
    its = []
    for t in range(steps):
        nat.mpc_run_ptr(Bn, 1, x.data_ptr(), xbar.data_ptr(), e.data_ptr(), noise[t].data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), st.data_ptr())
        nat.sync(); its.append(nat.last_iterations(Bn).mean())
    print(case, "policy", pol, "mean its per step:", " ".join(f"{v:.1f}" for v in its), "| cost mean", float(cost.mean()), "x0 mean", float(x[:,0].mean()))
