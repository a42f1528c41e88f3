import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
import bench
from tzddpc_amd.dist import vertex_noise
ctl, A, Bm, zon = bench.build_controller(0, 20)
nat = ctl._native
B, T = 1024, 30
dev = torch.device("cuda", 0)
Wv = zon.W.compute_vertices()
noise = torch.from_numpy(vertex_noise(Wv, 0, B, T)).to(dev)            # B x T x n
x0 = torch.from_numpy(np.tile(zon.X0.center, (B, 1))).to(dev)
At = torch.from_numpy(np.ascontiguousarray(A)).to(dev); Bt = torch.from_numpy(np.ascontiguousarray(Bm)).to(dev)
xt = torch.empty((B, T + 1, 2), dtype=torch.float64, device=dev); ut = torch.empty((B, T, 1), dtype=torch.float64, device=dev)
cost = torch.empty((B, T), dtype=torch.float64, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    nat.simulate_batch_ptr(B, T, x0.data_ptr(), noise.data_ptr(), At.data_ptr(), Bt.data_ptr(), xt.data_ptr(), ut.data_ptr(), cost.data_ptr(), st.data_ptr())
    nat.sync()
    dt = time.perf_counter() - t0
    print(f"simulate_batch (C loop, no timing events): {dt/T*1e3:.3f} ms/step  {B*T/dt:.0f} steps/s  bad={int((st!=0).sum())}")
nat.timing_enable(True)
t0 = time.perf_counter()
nat.simulate_batch_ptr(B, T, x0.data_ptr(), noise.data_ptr(), At.data_ptr(), Bt.data_ptr(), xt.data_ptr(), ut.data_ptr(), cost.data_ptr(), st.data_ptr())
nat.sync(); dt = time.perf_counter() - t0
print(f"with timing events: {dt/T*1e3:.3f} ms/step; kernel ms/step:", [nat.timing_get(k)[0]/T for k in range(4)])
