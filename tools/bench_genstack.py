"""K1g bench: literal stacked-generator tubes (tz_genstack_intervals) -- python tools/bench_genstack.py [system] [N] [k0] [B] [reps]
One JSON line: tube evaluations / s, the streaming kernel's time (HIP events inside the library), both rooflines -- HBM (the stack
is read once per tile of 256 trajectories: 8 n (1 + n + m) bytes per generator and tile) and f64 FMA throughput (2 n (n + m) + 2 n +
2 m n flop per generator and trajectory) -- and which one binds; checked against the numpy statement of the same sums."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from tzddpc_amd import native
from tzddpc_amd.genstack import build_stack, evaluate_host
from tzddpc_amd import TZDDPC, Data
from tzddpc_amd.harness import generate_trajectories, system

name = sys.argv[1] if len(sys.argv) > 1 else "dim5_w001"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
k0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 11
k0 = None if k0 < 0 else k0
A, Bm, zon, T = system(name)
n, m = Bm.shape
ctl = TZDDPC(generate_trajectories(A, Bm, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(25)))
ctl.build_zonotopes_theta(zon)
t0 = time.time(); st = build_stack(ctl.MdataK, ctl.Mdelta, ctl.theta.K, zon.W, n, m, N, k0, nseg=N); t_build = time.time() - t0
gs = native.GenStack(0, st)
info = gs.info()
rng = np.random.default_rng(3)
e0 = 0.02 * rng.standard_normal((B, n)); zeta = rng.standard_normal((B, N, n + m))
dev = torch.device("cuda", 0)
te, tz = torch.from_numpy(e0).to(dev), torch.from_numpy(zeta).to(dev)
c = torch.empty((B, N, n), dtype=torch.float64, device=dev); rx = torch.empty_like(c); ru = torch.empty((B, N, m), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
ms = [gs.intervals_ptr(B, te.data_ptr(), tz.data_ptr(), c.data_ptr(), rx.data_ptr(), ru.data_ptr()) for _ in range(reps + 2)][2:]
torch.cuda.synchronize()
ref = evaluate_host(st, e0[7], zeta[7])
err = max(np.abs(rx[7].cpu().numpy() - np.array([r[1] for r in ref])).max() / (1 + max(r[1].max() for r in ref)),
          np.abs(c[7].cpu().numpy() - np.array([r[0] for r in ref])).max())
med = float(np.median(ms))
tiles = (B + 255) // 256
bytes_alg = info["stack_bytes"] * tiles + B * (n + N * (n + m)) * 8 + info["chunks"] * B * (n + m) * 8
flop = info["generators"] * B * (2 * n * (n + m) + 2 * n + 2 * m * n + n + m)
gbs, tfs = bytes_alg / (med * 1e-3) / 1e9, flop / (med * 1e-3) / 1e12
line = {"metric": "literal tube evaluations/s (K1g, tz_genstack_intervals)", "value": B * N / (med * 1e-3), "unit": "tube intervals/s",
        "config": {"workload": f"{name} n={n} m={m}, N={N}, k0={k0}: literal Ze[0..N-1] of the reference's generator stacking, {info['generators']} generators "
                               f"({info['stack_bytes'] / 1e6:.1f} MB), {B} trajectories", "generators": info["generators"], "max_generators_per_tube": int(st.num_generators.max()),
                   "chunks": info["chunks"], "trajectories": B, "stack_build_s": t_build},
        "dtype": "f64", "data": "synthetic", "kernel_ms_median": med, "kernel_ms_all": [round(v, 4) for v in ms],
        "roofline": {"kernel": "tz_genstack_kernel", "hbm": {"achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "algorithmic_bytes": bytes_alg},
                     "f64_fma": {"achieved": tfs, "peak": 78.6, "unit": "TFLOP/s", "frac": tfs / 78.6, "algorithmic_flop": flop},
                     "bound": "hbm" if gbs / 8000.0 > tfs / 78.6 else "f64 vector FMA (one stack pass serves 256 trajectories: 73 flop per byte)"},
        "max_err_vs_numpy": float(err)}
print(json.dumps(line))
