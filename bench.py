#!/usr/bin/env python
"""bench.py -- MPC steps/s of the TZDDPC hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): double integrator (n=2, m=1), horizon N=20,
full build_problem, complexity-script zonotopes, 1024 closed-loop trajectories PER GPU (weak scaling),
vertex-of-W noise with PCG64(1000 + global trajectory index).  One "step" = one MPC step of every trajectory of the
rank: tube propagation + parameter application + interior-point QP solve + recovery + plant update
(tz_mpc_step), state resident in HBM.  Trajectories are independent: ranks share nothing on the data path; one
all-gather of (cost, final state) per trajectory closes the timed region (RCCL over xGMI).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (== vector) peak, AMD spec; tools/mfma_f64_rate.hip measures 73 on v_mfma_f64_4x4x4


def di_loss(u, x):
    from tzddpc_amd import cplite as cp
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(x[i, :], p=2) ** 2 + 1e-2 * cp.norm(u[i], p=1)
    return cost


def build_controller(device, horizon):
    from tzddpc_amd import TZDDPC
    from tzddpc_amd.harness import generate_trajectories, system
    A, B, zon, T = system("di_cc")
    rng = np.random.default_rng(25)
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, rng), device=device)
    ctl.build_zonotopes_theta(zon)
    kw = {"tol": float(os.environ["TZ_TOL"])} if "TZ_TOL" in os.environ else {}
    if "TZ_STEP_FRAC" in os.environ:
        kw["step_frac"] = float(os.environ["TZ_STEP_FRAC"])
    ctl.build_problem(horizon, di_loss, lambda u, x: [], **kw)
    return ctl, A, B, zon


def cpu_baseline(ctl, A, B, zon, horizon, warmup, steps, seconds_budget=40.0):
    """Plain-C oracle (oracle/c/tz_oracle.c, same algorithm incl. the closed-loop warm start) on the host cores: the same
    closed-loop workload on a bounded number of trajectories; the `warmup` leading steps are timed separately and subtracted,
    so the rate covers the same steps as the GPU number."""
    from oracle.c_oracle import COracle
    from tzddpc_amd.dist import vertex_noise
    from tzddpc_amd.builder import horizon_shift
    pol = int(ctl.warm_shift_policy)       # same warm-start policy as the device chose at build time (the shift maps are data handed to the oracle)
    co = COracle(ctl.qp, shift_policy=pol, shift_maps=horizon_shift(ctl.qp) if pol else None)
    cores = max(1, min(os.cpu_count() or 1, COracle.max_threads(), 16))
    Wv = zon.W.compute_vertices()
    T = warmup + steps

    def timed(traj, t):
        x0 = np.tile(zon.X0.center, (traj, 1))
        t0 = time.perf_counter(); out = co.simulate_batch(x0, vertex_noise(Wv, 0, traj, T)[:, :t], A, B, threads=cores)
        return time.perf_counter() - t0, out

    dt, _ = timed(cores, T)                                                  # calibration: one trajectory per thread
    traj = int(max(cores, min(16384, cores * max(1, int(seconds_budget / max(dt * (1.0 + warmup / T), 1e-9))))))
    d_all, out = timed(traj, T)
    d_warm = timed(traj, warmup)[0] if warmup > 0 else 0.0
    dt = max(d_all - d_warm, 1e-9)
    # one thread, for scale (SURVEY section 8d asks for both)
    t1 = max(32, traj // (8 * cores))
    x1 = np.tile(zon.X0.center, (t1, 1)); n1 = vertex_noise(Wv, 0, t1, T)
    t0 = time.perf_counter(); co.simulate_batch(x1, n1, A, B, threads=1); e_all = time.perf_counter() - t0
    t0 = time.perf_counter(); co.simulate_batch(x1, n1[:, :warmup], A, B, threads=1); e_warm = (time.perf_counter() - t0) if warmup > 0 else 0.0
    one = t1 * steps / max(e_all - e_warm, 1e-9)
    return {"value": traj * steps / dt, "unit": "MPC steps/s", "cores": cores, "kind": "port", "value_one_thread": one,
            "sample": f"{traj} trajectories x closed-loop steps {warmup}..{T - 1} of the same DI N={horizon} workload (time of {T} steps minus time of the "
                      f"first {warmup}), plain-C oracle with the same warm-started interior point, OpenMP over trajectories, {d_all + d_warm:.1f} s wall, "
                      f"all statuses zero: {bool((out['status'] == 0).all())}"}


def ensure_built():
    """The in-tree libraries normally travel with the snapshot; on a bare checkout build them once (one rank at a time)."""
    import fcntl
    lib = os.path.join(ROOT, "tzddpc_amd", "lib", "libtzddpc_hip.so")
    orc = os.path.join(ROOT, "oracle", "_build", "libtz_oracle.so")
    if os.path.exists(lib) and os.path.exists(orc):
        return
    os.makedirs(os.path.join(ROOT, "tzddpc_amd", "lib"), exist_ok=True)
    with open(os.path.join(ROOT, "tzddpc_amd", "lib", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            import __graft_entry__
            __graft_entry__.build()
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="trajectories per GPU")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    ensure_built()
    import torch
    import torch.distributed as dist
    from tzddpc_amd.dist import gather_results, shard_range, vertex_noise

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ            # under torch.distributed.run even one rank goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    ctl, A, Bm, zon = build_controller(local_rank, args.horizon)
    nat = ctl._native
    # one HIP stream for the kernel, torch's copies and the collective: the exchange is ordered after the closed loop on the device,
    # without a host round trip in between
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    nat.set_stream(side.cuda_stream)
    n, m = ctl.qp.n, ctl.qp.m
    Bl = args.batch
    total = Bl * world
    lo, hi = shard_range(total, world, rank)
    assert hi - lo == Bl
    K, W = args.steps, args.warmup
    Wv = zon.W.compute_vertices()
    noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(Wv, lo, Bl, K + W).transpose(1, 0, 2))).to(dev)   # (K+W) x B x n
    x = torch.from_numpy(np.tile(zon.X0.center, (Bl, 1))).to(dev)
    xbar = x.clone(); e = torch.zeros_like(x)
    u = torch.zeros((Bl, m), dtype=torch.float64, device=dev)
    cost = torch.zeros(Bl, dtype=torch.float64, device=dev)
    status = torch.zeros(Bl, dtype=torch.int32, device=dev)
    bad = torch.zeros(Bl, dtype=torch.int32, device=dev)
    At = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev)
    Bt = torch.from_numpy(np.ascontiguousarray(Bm, dtype=np.float64)).to(dev)
    torch.cuda.synchronize()

    ptrs = (x.data_ptr(), xbar.data_ptr(), e.data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), status.data_ptr())
    wptr = [noise[0].data_ptr(), noise[W].data_ptr()] if W > 0 else [noise[0].data_ptr()] * 2
    res = torch.empty((Bl, 1 + n), dtype=torch.float64, device=dev)          # per-trajectory [cost | final state]: what the ranks exchange

    def run(first, k):   # k closed-loop steps, issued by one C call (tz_mpc_run): no host work between steps
        nat.mpc_run_ptr(Bl, k, ptrs[0], ptrs[1], ptrs[2], wptr[0] if first else wptr[1], ptrs[3], ptrs[4], ptrs[5], ptrs[6], ptrs[7])

    def collect():
        res[:, 0] = cost; res[:, 1:] = x
        return gather_results(res, total)

    if W > 0:
        run(True, W)
    nat.sync()
    bad |= (status != 0).int()
    _ = collect()                                                       # warm torch's copy / RCCL paths outside the timed region
    nat.timing_enable(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(); nat.sync()
    t0 = time.perf_counter()
    run(False, K)
    ta = time.perf_counter()
    gathered = collect()                                                      # per-trajectory cost + final state only
    tb = time.perf_counter()
    torch.cuda.synchronize()
    tc = time.perf_counter()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("TZ_BENCH_DEBUG"):
        print(f"[rank {rank}] launch {1e3*(ta-t0):.3f} collect (issue) {1e3*(tb-ta):.3f} synchronize {1e3*(tc-tb):.3f} barrier {1e3*(elapsed-(tc-t0)):.3f} ms", file=sys.stderr)
    ipm_ms, ipm_n = nat.timing_get(1)
    prep_ms, _ = nat.timing_get(0); fin_ms, _ = nat.timing_get(2); plant_ms, _ = nat.timing_get(3)
    bad |= (status != 0).int()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    nbad = bad.sum().to(torch.float64).reshape(1)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(nbad, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())

    if rank == 0:
        # roofline of the dominant kernel (tz_ipm_kernel): useful MFMA flops of the static plan x iterations
        work = nat.work_get()                # counted on the device by tz_ipm_kernel during the timed launches
        fact_per_launch = work["factorizations"] / max(ipm_n, 1)
        iters_mean = work["factorizations"] / max(work["trajectory_solves"], 1)
        flop_per_launch = nat.alg_flops["per_factorization"] * fact_per_launch
        avg_ms = ipm_ms / max(ipm_n, 1)
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12
        line = {
            "metric": "MPC steps/sec (batched trajectories), double-integrator N=20",
            "value": total * K / elapsed, "unit": "MPC steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"double integrator n=2 m=1, horizon N={args.horizon}, full build_problem, {Bl} closed-loop trajectories per GPU "
                                   f"(BASELINE.json configs[1]), complexity-script zonotopes, vertex-of-W noise PCG64(1000+i)",
                       "trajectories_per_gpu": Bl, "horizon": args.horizon, "nz": ctl.qp.nz, "rows": int(nat.mi),
                       "ipm_factorizations_per_trajectory_step": iters_mean, "warm_start": os.environ.get("TZ_WARM", "1") != "0", "warm_shift_policy": int(ctl.warm_shift_policy), "unsolved_trajectory_steps": int(nbad.item()),
                       "kernel_ms_per_step": {"tz_tube+affine": prep_ms / K, "tz_ipm": ipm_ms / K, "tz_finish": fin_ms / K, "tz_plant": plant_ms / K},
                       "steps_per_launch": K / max(ipm_n, 1)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F64_MFMA_PEAK_TFLOPS, "traffic": None,
                         "kernel": "tz_ipm_kernel", "avg_launch_ms": avg_ms, "launches": int(ipm_n),
                         "flop_per_launch": flop_per_launch,
                         "flop_per_factorization": nat.alg_flops["per_factorization"],
                         "dense_flop_per_factorization": nat.alg_flops["dense_per_factorization"],
                         "note": "algorithmic f64 flops of one interior-point factorisation counted on the non-zeros of G (sparse outer products "
                                 "for G'WG, Cholesky, four G/G' products, two solve pairs, P x) x factorisations counted on the device; the dense "
                                 "count is given beside it; prologue / recovery / plant work of the fused step is not counted"},
        }
        line["config"]["gathered_rows"] = int(gathered.shape[0])
        try:      # HBM-side bytes of the timed launch as measured by the PMC passes committed under profiles/ (same steps and batch only)
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            if tr["steps"] == K and tr["trajectories_per_gpu"] == Bl and ipm_n == 1:
                line["roofline"]["traffic"] = tr["fetch_bytes"] + tr["write_bytes"]
                line["roofline"]["traffic_source"] = tr["source"]
        except Exception:
            pass
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(ctl, A, Bm, zon, args.horizon, W, K)
            except Exception as ex:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "MPC steps/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
