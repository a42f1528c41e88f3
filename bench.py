#!/usr/bin/env python
"""bench.py -- MPC steps/s of the TZDDPC hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config NAME] [--repeats R]

Default workload (BASELINE.json configs[1]): double integrator (n=2, m=1), horizon N=20,
full build_problem, complexity-script zonotopes, 1024 closed-loop trajectories PER GPU (weak scaling), vertex-of-W noise with
PCG64(1000 + global trajectory index).  One "step" = one MPC step of every trajectory of the rank: tube propagation + parameter
application + interior-point QP solve + recovery + plant update, state resident in HBM, all K steps in one launch (tz_mpc_run).
Trajectories are independent: ranks share nothing on the data path; one all-gather of (cost, final state) per trajectory closes
the timed region (RCCL over xGMI).

The timed window (K steps after W untimed warm-up steps from X0, barrier + synchronize on both sides, max over ranks) is
repeated `--repeats` times, every time from a fresh start (state back to X0, warm-start state of the handle reset, the W warm-up
steps run again untimed): `value` is the MEDIAN window, the spread is reported beside it.

--config selects the other BASELINE configurations: pulley_n10 (config 3), dim5_n20 (config 4),
di_n5 / di_n10 / di_n40 / di_n80 and the simplified di_n20_k1 / di_n20_k2 (config 5).
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


def _loss_di(u, x):            # reference examples/1.double_integrator_sim.py:22-28
    from tzddpc_amd import cplite as cp
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(x[i, :], p=2) ** 2 + 1e-2 * cp.norm(u[i], p=1)
    return cost


def _loss_pulley(u, y):        # reference examples/2.pulley_sim.py:17-22
    from tzddpc_amd import cplite as cp
    cost = 0
    for i in range(u.shape[0]):
        cost += cp.norm(y[i, 0] - 1, p=2)
    return cost


def _loss_dim5(u, x):          # reference examples/3.5dimsystem_sim.py:14-20
    from tzddpc_amd import cplite as cp
    cost = 0
    for i in range(u.shape[0]):
        cost += 1e9 * cp.norm(x[i, 1] - 2, p=2) + 1e-1 * cp.norm(u[i], p=2)
    return cost


def _cons_dim5(u, x):          # reference examples/3.5dimsystem_sim.py:23-26
    return [x[:, 1] <= 10, x[:, 1] >= 2]


def _nocons(u, x):
    return []


# name: (system, loss, constraints, horizon, k0, default trajectories per GPU, description)
CONFIGS = {
    "di_n20": ("di_cc", _loss_di, _nocons, 20, None, 1024, "double integrator n=2 m=1, horizon N=20, full build_problem (BASELINE.json configs[1])"),
    "pulley_n10": ("pulley", _loss_pulley, _nocons, 10, None, 4096, "pulley n=4 m=1, horizon N=10, L1 tracking loss (BASELINE.json configs[2])"),
    "dim5_n20": ("dim5_w001", _loss_dim5, _cons_dim5, 20, None, 1024, "5-dim system n=5 m=1 (the reference example's m), horizon N=20, W scaled to 0.01 (BASELINE.json configs[3])"),
    "di_n5": ("di_cc", _loss_di, _nocons, 5, None, 2048, "double integrator, horizon sweep N=5 (BASELINE.json configs[4])"),
    "di_n10": ("di_cc", _loss_di, _nocons, 10, None, 2048, "double integrator, horizon sweep N=10 (BASELINE.json configs[4])"),
    "di_n40": ("di_cc", _loss_di, _nocons, 40, None, 1024, "double integrator, horizon sweep N=40 (BASELINE.json configs[4])"),
    "di_n80": ("di_cc", _loss_di, _nocons, 80, None, 1024, "double integrator, horizon sweep N=80 (BASELINE.json configs[4])"),
    "di_n20_k1": ("di_cc", _loss_di, _nocons, 20, 1, 1024, "double integrator N=20, build_problem_simplified(k0=1) (BASELINE.json configs[4])"),
    "di_n20_k2": ("di_cc", _loss_di, _nocons, 20, 2, 1024, "double integrator N=20, build_problem_simplified(k0=2) (BASELINE.json configs[4])"),
}


def build_controller(device, config, horizon=None):
    from tzddpc_amd import TZDDPC
    from tzddpc_amd.harness import generate_trajectories, system
    sysname, loss, cons, N, k0, _, _ = CONFIGS[config]
    N = int(horizon) if horizon else N
    A, B, zon, T = system(sysname)
    rng = np.random.default_rng(25)
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, rng), device=device)
    ctl.build_zonotopes_theta(zon)
    kw = {"tol": float(os.environ["TZ_TOL"])} if "TZ_TOL" in os.environ else {}
    if "TZ_STEP_FRAC" in os.environ:
        kw["step_frac"] = float(os.environ["TZ_STEP_FRAC"])
    if k0 is None:
        ctl.build_problem(N, loss, cons, **kw)
    else:
        ctl.build_problem_simplified(k0, N, loss, cons, **kw)
    return ctl, A, B, zon, N


def cpu_baseline(ctl, A, B, zon, label, warmup, steps, seconds_budget=25.0):
    """Plain-C oracle (oracle/c/tz_oracle.c: the same algorithm incl. the closed-loop warm start, own scaling / Cholesky) on ALL host
    cores: the same closed-loop workload on a bounded number of trajectories; the `warmup` leading steps are timed separately and
    subtracted, so the rate covers the same steps as the GPU number."""
    from oracle.c_oracle import COracle
    from tzddpc_amd.builder import horizon_shift
    from tzddpc_amd.dist import vertex_noise
    pol = int(ctl.warm_shift_policy)       # same warm-start policy as the device chose at build time (the shift maps are data handed to the oracle)
    co = COracle(ctl.qp, shift_policy=pol, shift_maps=horizon_shift(ctl.qp) if pol else None, warm_gain=float(getattr(ctl, "warm_push_gain", 1.0)), warm_cap=float(getattr(ctl, "warm_push_cap", 1e300)), mu_factor=float(getattr(ctl, "mu_factor", 1e-3)))
    host_cpus = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host_cpus
    cores = max(1, min(usable, COracle.max_threads()))
    Wv = zon.W.compute_vertices()
    T = warmup + steps

    def timed(traj, t, threads):
        x0 = np.tile(zon.X0.center, (traj, 1))
        t0 = time.perf_counter(); out = co.simulate_batch(x0, vertex_noise(Wv, 0, traj, T)[:, :t], A, B, threads=threads)
        return time.perf_counter() - t0, out

    dt, _ = timed(cores, T, cores)                                           # calibration: one trajectory per thread
    traj = int(max(cores, min(16384, cores * max(1, int(0.6 * seconds_budget / max(dt * (1.0 + warmup / T), 1e-9))))))
    d_all, out = timed(traj, T, cores)
    d_warm = timed(traj, warmup, cores)[0] if warmup > 0 else 0.0
    dt = max(d_all - d_warm, 1e-9)
    # one thread, for scale (SURVEY section 8d asks for both)
    t1 = max(8, traj // (8 * cores))
    e_all = timed(t1, T, 1)[0]; e_warm = timed(t1, warmup, 1)[0] if warmup > 0 else 0.0
    one = t1 * steps / max(e_all - e_warm, 1e-9)
    return {"value": traj * steps / dt, "unit": "MPC steps/s", "cores": cores, "host_cpu_count": host_cpus, "usable_cpus": usable, "kind": "port",
            "value_one_thread": one,
            "reference_published": "~18 MPC steps/s: the reference's own pulley N=2 closed loop incl. build, 5 runs, hardware unstated "
                                   "(examples/results/pulley.tzddpc_times.npy; BASELINE.md) -- context only, the reference cannot run on this box",
            "sample": f"{traj} trajectories x closed-loop steps {warmup}..{T - 1} of the same {label} workload (time of {T} steps minus time of the "
                      f"first {warmup}), plain-C oracle with the same warm-started interior point, OpenMP over trajectories on {cores} threads "
                      f"(os.cpu_count() = {host_cpus}), {d_all + d_warm:.1f} s wall, all statuses zero: {bool((out['status'] == 0).all())}"}


def ensure_built():
    """Build (or confirm current: content hash of sources and flags) the in-tree libraries, one rank at a time."""
    import fcntl
    os.makedirs(os.path.join(ROOT, "tzddpc_amd", "lib"), exist_ok=True)
    with open(os.path.join(ROOT, "tzddpc_amd", "lib", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            import __graft_entry__
            __graft_entry__.build()
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def archived_traffic(config, nz, mi, Bl, K):
    """HBM-side bytes of one timed launch from the PMC passes committed under profiles/ (rocprofv3 --pmc, separate passes), kept
    per trajectory-step and scaled to this launch; None when the archive was taken on another problem shape."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        e = tr["configs"][config]
        if int(e["nz"]) != int(nz) or int(e["rows"]) != int(mi):
            return None, None
        import __graft_entry__
        same = e.get("lib_source_hash") == __graft_entry__.source_hash()
        per = float(e["fetch_bytes_per_trajectory_step"]) + float(e["write_bytes_per_trajectory_step"])
        return per * Bl * K, {"source": e["source"], "bytes_per_trajectory_step": per, "archived_at_steps": e["steps"],
                              "library_unchanged_since_archive": bool(same)}
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="di_n20", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (default: the configuration's)")
    ap.add_argument("--horizon", type=int, default=0, help="override the configuration's horizon")
    ap.add_argument("--repeats", type=int, default=11, help="timed windows (median reported)")
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

    ctl, A, Bm, zon, horizon = build_controller(local_rank, args.config, args.horizon)
    nat = ctl._native
    # one HIP stream for the kernel, torch's copies and the collective: the exchange is ordered after the closed loop on the device,
    # without a host round trip in between
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    nat.set_stream(side.cuda_stream)
    n, m = ctl.qp.n, ctl.qp.m
    Bl = args.batch or CONFIGS[args.config][5]
    total = Bl * world
    lo, hi = shard_range(total, world, rank)
    assert hi - lo == Bl
    K, W, R = args.steps, args.warmup, max(1, args.repeats)
    Wv = zon.W.compute_vertices()
    noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(Wv, lo, Bl, K + W).transpose(1, 0, 2))).to(dev)   # (K+W) x B x n
    x_init = torch.from_numpy(np.tile(zon.X0.center, (Bl, 1))).to(dev)
    x = x_init.clone(); xbar = x.clone(); e = torch.zeros_like(x)
    u = torch.zeros((Bl, m), dtype=torch.float64, device=dev)
    cost = torch.zeros(Bl, dtype=torch.float64, device=dev)
    status = torch.zeros(Bl, dtype=torch.int32, device=dev)
    bad = torch.zeros(Bl, dtype=torch.int32, device=dev)
    At = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev)
    Bt = torch.from_numpy(np.ascontiguousarray(Bm, dtype=np.float64).reshape(n, m)).to(dev)
    torch.cuda.synchronize()

    ptrs = (x.data_ptr(), xbar.data_ptr(), e.data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), status.data_ptr())
    wptr = [noise[0].data_ptr(), noise[W].data_ptr()] if W > 0 else [noise[0].data_ptr()] * 2
    res = torch.empty((Bl, 1 + n), dtype=torch.float64, device=dev)          # per-trajectory [cost | final state]: what the ranks exchange

    def run(first, k):   # k closed-loop steps, issued by one C call (tz_mpc_run): no host work between steps
        nat.mpc_run_ptr(Bl, k, ptrs[0], ptrs[1], ptrs[2], wptr[0] if first else wptr[1], ptrs[3], ptrs[4], ptrs[5], ptrs[6], ptrs[7])

    def collect():
        res[:, 0] = cost; res[:, 1:] = x
        return gather_results(res, total)

    def fresh_start():   # state back to X0, no memory of earlier solves in the handle, W untimed warm-up steps
        x.copy_(x_init); xbar.copy_(x_init); e.zero_()
        torch.cuda.synchronize()
        nat.reset_warm()
        if W > 0:
            run(True, W)
        nat.sync()
        return (status != 0).int()

    bad |= fresh_start()
    _ = collect()                                                       # warm torch's copy / RCCL paths outside the timed region
    windows, kern_ms, facts, solves = [], [], [], []
    gathered = None
    for rep in range(R):
        if rep > 0:
            bad |= fresh_start()
        nat.timing_enable(True)                                          # zeroes the event sums / work counters of the library
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(); nat.sync()
        t0 = time.perf_counter()
        run(False, K)
        gathered = collect()                                             # per-trajectory cost + final state only
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        windows.append(float(tmax.item()))
        ms, cnt = nat.timing_get(1)                                      # HIP events recorded by the library on its own stream around the launch
        work = nat.work_get()
        kern_ms.append(ms / max(cnt, 1)); facts.append(work["factorizations"]); solves.append(work["trajectory_solves"])
        bad |= (status != 0).int()
    nbad = bad.sum().to(torch.float64).reshape(1)
    if use_dist:
        dist.all_reduce(nbad, op=dist.ReduceOp.SUM)

    if rank == 0:
        order = np.argsort(windows)
        med = int(order[len(order) // 2])                                # the median window and ITS work counters
        elapsed = windows[med]
        fact_per_launch = facts[med]
        iters_mean = facts[med] / max(solves[med], 1)
        # every trajectory-step also evaluates the stopping test of its starting point once more than it factorises (a warm start
        # that is already optimal costs exactly this and no factorisation): residual products P x, G'lambda, G x
        flop_per_test = nat.alg_flops["p_products"] + 0.5 * nat.alg_flops["g_products"]
        flop_per_fixed = nat.alg_flops["per_step_fixed"]               # tube propagation (SURVEY 8d F_tube), affine maps, recovery, plant
        flop_per_launch = nat.alg_flops["per_factorization"] * fact_per_launch + (flop_per_test + flop_per_fixed) * solves[med]
        avg_ms = float(np.median(kern_ms))
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12
        sysname, _, _, _, k0, _, desc = CONFIGS[args.config]
        label = f"{args.config} (N={horizon})"
        line = {
            "metric": "MPC steps/sec (batched trajectories), double-integrator N=20" if args.config == "di_n20" and horizon == 20
                      else f"MPC steps/sec (batched trajectories), {label}",
            "value": total * K / elapsed, "unit": "MPC steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{desc}; {Bl} closed-loop trajectories per GPU, {'complexity-script' if sysname == 'di_cc' else 'example'} zonotopes, "
                                   f"vertex-of-W noise PCG64(1000+i), all {K} timed steps in one launch",
                       "name": args.config, "trajectories_per_gpu": Bl, "horizon": horizon, "k0": k0, "nz": ctl.qp.nz, "rows": int(nat.mi),
                       "ipm_factorizations_per_trajectory_step": iters_mean, "warm_start": os.environ.get("TZ_WARM", "1") != "0",
                       "warm_shift_policy": int(ctl.warm_shift_policy), "warm_push_gain": float(ctl.warm_push_gain), "warm_push_cap": (None if not np.isfinite(ctl.warm_push_cap) else float(ctl.warm_push_cap)), "mu_factor": float(ctl.mu_factor), "unsolved_trajectory_steps": int(nbad.item()),
                       "gathered_rows": int(gathered.shape[0]), "lds_bytes_per_workgroup": nat.plan_info()["lds_bytes"]},
            "timing": {"repeats": R, "reported": "median window", "window_ms": [round(w * 1e3, 4) for w in windows],
                       "window_ms_min": min(windows) * 1e3, "window_ms_max": max(windows) * 1e3,
                       "spread_rel": (max(windows) - min(windows)) / elapsed,
                       "value_best": total * K / min(windows), "value_worst": total * K / max(windows),
                       "fresh_start_per_window": "state reset to X0, tz_problem_reset_warm, warm-up steps re-run untimed"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F64_MFMA_PEAK_TFLOPS, "traffic": None,
                         "kernel": "tz_ipm_kernel", "avg_launch_ms": avg_ms, "launch_ms_all": [round(v, 5) for v in kern_ms], "launches": R,
                         "flop_per_launch": flop_per_launch,
                         "flop_per_factorization": nat.alg_flops["per_factorization"],
                         "flop_per_stopping_test": flop_per_test, "flop_per_step_fixed_part": flop_per_fixed, "factorizations_per_launch": fact_per_launch, "trajectory_steps_per_launch": solves[med],
                         "dense_flop_per_factorization": nat.alg_flops["dense_per_factorization"],
                         "note": "algorithmic f64 flops of one interior-point factorisation counted on the non-zeros of G (sparse outer products "
                                 "for G'WG, Cholesky, four G/G' products, two solve pairs, P x) x factorisations counted on the device, plus one "
                                 "stopping test (P x, G'lambda, G x) per trajectory-step -- a warm start that is already optimal costs only that; "
                                 "and the fixed part of a step (tube propagation F_tube of SURVEY 8d, the affine maps over theta, recovery, plant update); "
                                 "the dense count of a factorisation is given beside it; launch time = "
                                 "median over the repeats of the HIP-event time of the one timed launch"},
        }
        tb, tinfo = archived_traffic(args.config, ctl.qp.nz, nat.mi, Bl, K)
        if tb is not None:
            line["roofline"]["traffic"] = tb
            line["roofline"]["traffic_source"] = tinfo
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(ctl, A, Bm, zon, label, W, K)
            except Exception as ex:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "MPC steps/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
