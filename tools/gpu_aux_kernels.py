"""One pass over the auxiliary kernels (K0 tz_identify_kernel, tz_specrad_kernel, tz_adversary_kernel) at the sizes the reference uses,
for `rocprofv3 --kernel-trace --stats` (tools/profile_aux.sh): python tools/gpu_aux_kernels.py [repeats]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from tzddpc_amd import native
from tzddpc_amd.gain import lqr_gain
from tzddpc_amd.harness import generate_trajectories, system
from tzddpc_amd.zonotope import compute_LTI_matrix_zonotope, concatenate_zonotope
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, nsets in (("di_cc", 256), ("pulley", 256), ("dim5_w001", 256), ("dim5m2_w001", 64)):
    A, B, zon, T = system(name)
    n, m = B.shape
    us, xs = [], []
    for sd in range(nsets):
        d = generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(100 + sd))
        us.append(np.asarray(d.u)); xs.append(np.asarray(d.x))
    us, xs = np.stack(us), np.stack(xs)
    first = native.identify_batch(0, us, xs, zon.W.center)
    Ks = np.stack([lqr_gain(first["C"][b][:, :n], first["C"][b][:, n:]) for b in range(nsets)])
    t0 = time.perf_counter()
    for _ in range(reps):
        out = native.identify_batch(0, us, xs, zon.W.center, K=Ks)
    dt = (time.perf_counter() - t0) / reps
    print(f"K0 {name}: {nsets} data sets x {T} samples (n={n}, m={m}): {dt * 1e3:.3f} ms per call incl. host copies, all ok {bool((out['status'] == 0).all())}")
    # gain synthesis kernels on the un-reduced Mdata of the first data set (reference tzddpc/utils.py:105-129: 1146 samples; :13-41: 10 starts)
    Mw = concatenate_zonotope(zon.W, us.shape[1] - 1)
    Md = compute_LTI_matrix_zonotope(xs[0][:-1], xs[0][1:], us[0][:-1], Mw)
    K = Ks[0]
    IK = np.vstack([np.eye(n), K])
    M0 = Md.center @ IK
    Hg = np.stack([G @ IK for G in Md.generators])
    rng = np.random.default_rng(3)
    for S in (1146, 16384):
        beta = rng.uniform(-1, 1, size=(S, Hg.shape[0]))
        t0 = time.perf_counter()
        for _ in range(reps):
            rho, st = native.specrad_batch(0, M0, Hg, beta)
        print(f"specrad {name}: {S} samples x {Hg.shape[0]} generators: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call incl. host copies, max rho {rho.max():.4f}, ok {bool((st == 0).all())}")
    for S in (10, 1024):
        beta0 = rng.uniform(-1, 1, size=(S, Hg.shape[0]))
        t0 = time.perf_counter()
        for _ in range(reps):
            beta, fro, steps = native.adversary_batch(0, M0, Hg, beta0)
        print(f"adversary {name}: {S} starts x {Hg.shape[0]} generators: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call incl. host copies, max ||.||_F {fro.max():.4f}, steps <= {steps.max()}")
