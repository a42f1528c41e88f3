"""Host-side cost around the one-launch closed loop of the bench workload: wall time of tz_mpc_run + sync against the event time of the kernel.
python tools/gpu_host_overhead.py [steps]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
from tests import common
from tzddpc_amd.dist import vertex_noise
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
Bn, W = 1024, 5
ctl, (A, B, zon) = common.gpu_controller("di_n20")
nat = ctl._native; n, m = ctl.qp.n, ctl.qp.m
dev = torch.device("cuda", 0)
noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(zon.W.compute_vertices(), 0, Bn, K + W).transpose(1, 0, 2))).to(dev)
u = torch.zeros((Bn, m), dtype=torch.float64, device=dev); cost = torch.zeros(Bn, dtype=torch.float64, device=dev)
st = torch.zeros(Bn, dtype=torch.int32, device=dev)
At = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev); Bt = torch.from_numpy(np.ascontiguousarray(B, dtype=np.float64).reshape(n, m)).to(dev)
for timing in (False, True):
    walls, kers, calls = [], [], []
    for rep in range(6):
        x = torch.from_numpy(np.tile(zon.X0.center, (Bn, 1))).to(dev); xbar = x.clone(); e = torch.zeros_like(x)
        args = lambda t0: (x.data_ptr(), xbar.data_ptr(), e.data_ptr(), noise[t0].data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), st.data_ptr())
        nat.timing_enable(False)
        nat.mpc_run_ptr(Bn, W, *args(0)); nat.sync(); torch.cuda.synchronize()
        nat.timing_enable(timing)
        t0 = time.perf_counter(); nat.mpc_run_ptr(Bn, K, *args(W)); t1 = time.perf_counter(); nat.sync(); t2 = time.perf_counter()
        walls.append((t2 - t0) * 1e3); calls.append((t1 - t0) * 1e3)
        if timing: kers.append(nat.timing_get(1)[0])
    print(f"timing={timing}: wall ms {np.round(walls, 3)}, call-return ms {np.round(calls, 3)}, kernel event ms {np.round(kers, 3)}")
