#!/usr/bin/env python
"""bench.py -- MPC steps/s of the TZDDPC hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config NAME] [--repeats R]
  python bench.py -m TZDDPC|STZDDPC -ho HORIZON [-k0 K0] [-n NEVALS]     (flags of the reference's complexity harness,
                                                                           examples/1.double_integrator_computation_complexity.py:179-189)

--gpus N > 1 without a torchrun environment: this process starts N ranks itself (python -m torch.distributed.run, one per GPU)
BEFORE anything touches the GPU and relays rank 0's JSON line; under torchrun (RANK set) WORLD_SIZE must equal --gpus.

Default workload (BASELINE.json configs[1]): double integrator (n=2, m=1), horizon N=20,
full build_problem, complexity-script zonotopes, 1024 closed-loop trajectories PER GPU (weak scaling), vertex-of-W noise with
PCG64(1000 + global trajectory index).  One "step" = one MPC step of every trajectory of the rank: tube propagation + parameter
application + interior-point QP solve + recovery + plant update, state resident in HBM, all K steps in one launch (tz_mpc_run).
Trajectories are independent: ranks share nothing on the data path; one all-gather of (cost, final state) per trajectory closes
the timed region (RCCL over xGMI).

The timed window (K steps after W untimed warm-up steps from X0, barrier + synchronize on both sides, max over ranks) is
repeated `--repeats` times, every time from a fresh start (state back to X0, warm-start state of the handle reset, the W warm-up
steps run again untimed): `value` is the MEDIAN window, the spread is reported beside it.

--config selects the other BASELINE configurations: pulley_n10 (config 3), dim5_n20 (config 4, the reference
example's single input) and dim5m2_n20 (config 4 as BASELINE.json states it: two inputs), di_n5 / di_n10 / di_n40 / di_n80 and the
simplified di_n20_k1 / di_n20_k2 (config 5); genstack_dim5_k1 times kernel K1g (literal stacked-generator tubes).

Beside the driver-contract window the line carries `full_run` (SURVEY 8d's metric as written: T_sim = 50 closed-loop steps from X0,
nothing untimed) and `jittered_start` (every trajectory from its own point of X0 +- 0.25, seed 7).
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
    "dim5m2_n20": ("dim5m2_w001", _loss_dim5, _cons_dim5, 20, None, 1024, "5-dim system n=5 m=2 (BASELINE.json configs[3] as stated; second input column (1,0,1,0,1)', SURVEY 8d), horizon N=20, W scaled to 0.01"),
    "di_n5": ("di_cc", _loss_di, _nocons, 5, None, 2048, "double integrator, horizon sweep N=5 (BASELINE.json configs[4])"),
    "di_n10": ("di_cc", _loss_di, _nocons, 10, None, 2048, "double integrator, horizon sweep N=10 (BASELINE.json configs[4])"),
    "di_n40": ("di_cc", _loss_di, _nocons, 40, None, 1024, "double integrator, horizon sweep N=40 (BASELINE.json configs[4])"),
    "di_n80": ("di_cc", _loss_di, _nocons, 80, None, 1024, "double integrator, horizon sweep N=80 (BASELINE.json configs[4])"),
    "di_n20_k1": ("di_cc", _loss_di, _nocons, 20, 1, 1024, "double integrator N=20, build_problem_simplified(k0=1) (BASELINE.json configs[4])"),
    "di_n20_k2": ("di_cc", _loss_di, _nocons, 20, 2, 1024, "double integrator N=20, build_problem_simplified(k0=2) (BASELINE.json configs[4])"),
}


GENSTACK_CONFIGS = {
    # name: (system, horizon, k0, default trajectories, description)
    "genstack_dim5_k1": ("dim5_w001", 20, 1, 1024, "K1g: literal tubes Ze[0..19] of build_problem_simplified(k0=1) for the 5-dim system, 600 148 generators"),
    "genstack_pulley_k1": ("pulley", 20, 1, 1024, "K1g: literal tubes Ze[0..19] of build_problem_simplified(k0=1) for the pulley, 168 434 generators"),
}


def build_controller(device, config, horizon=None, k0_override="keep"):
    from tzddpc_amd import TZDDPC
    from tzddpc_amd.harness import generate_trajectories, system
    sysname, loss, cons, N, k0, _, _ = CONFIGS[config]
    N = int(horizon) if horizon else N
    if k0_override != "keep":
        k0 = k0_override
    A, B, zon, T = system(sysname)
    rng = np.random.default_rng(25)
    ctl = TZDDPC(generate_trajectories(A, B, zon.X0, zon.U, zon.W, 1, T, rng), device=device)
    ctl.build_zonotopes_theta(zon)
    if k0 is None:
        ctl.build_problem(N, loss, cons)
    else:
        ctl.build_problem_simplified(k0, N, loss, cons)
    return ctl, A, B, zon, N, k0


def cpu_budget():
    """What the host really gives this process: logical CPUs, affinity, the cgroup CPU quota (v2 cpu.max / v1 cfs_quota) and SMT
    siblings -- `effective_cpus` = min(affinity, quota) is the number of threads that can run at once."""
    host = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    smt = 1
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = max(1, len([p for part in sib.split(",") for p in ([part] if "-" not in part else range(int(part.split("-")[0]), int(part.split("-")[1]) + 1))]))
    except (OSError, ValueError):
        pass
    eff = float(usable) if quota is None else min(float(usable), quota)
    return {"host_cpu_count": host, "usable_cpus": usable, "cgroup_cpu_quota": quota, "smt_threads_per_core": smt, "effective_cpus": eff}


def _throttled():
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            if ln.startswith("nr_throttled"):
                return int(ln.split()[1])
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(ctl, A, B, zon, label, warmup, steps, full_steps=0, seconds_budget=40.0, repeats=3):
    """Plain-C oracle (oracle/c/tz_oracle.c: the same algorithm incl. the closed-loop warm start, own scaling / Cholesky; OpenMP over
    trajectories, every thread allocates and first-touches its own work area) on the host cores, same closed-loop workload on a
    bounded number of trajectories; the `warmup` leading steps are timed separately and subtracted, so the rate covers the same steps
    as the GPU number.  The host's real CPU budget is reported (`cpu_budget`), a thread-scaling curve 1, 2, 4, ... is measured on
    samples of ~2 s each (two halves, the better one kept), and `value` is the BEST point of the curve (median of `repeats` samples there), `cores` its thread count.
    `full_run`: SURVEY 8d's metric as written (T_sim steps from X0, nothing subtracted) at the same thread count."""
    from oracle.c_oracle import COracle
    from tzddpc_amd.builder import horizon_shift
    from tzddpc_amd.dist import vertex_noise
    pol = int(ctl.warm_shift_policy)       # same warm-start policy as the device chose at build time (the shift maps are data handed to the oracle)
    co = COracle(ctl.qp, shift_policy=pol, shift_maps=horizon_shift(ctl.qp) if pol else None, warm_gain=float(getattr(ctl, "warm_push_gain", 1.0)), warm_cap=float(getattr(ctl, "warm_push_cap", 1e300)), mu_factor=float(getattr(ctl, "mu_factor", 1e-3)),
                 stored_start=(None if getattr(ctl, "stored_start", None) is None else ctl.stored_start[0]))
    budget = cpu_budget()
    max_thr = max(1, min(budget["usable_cpus"], COracle.max_threads()))
    Wv = zon.W.compute_vertices()
    T = warmup + steps
    Tn = max(T, full_steps)
    t_wall0 = time.perf_counter()
    thr0 = _throttled()
    ok = [True]

    def timed(traj, t, threads):
        x0 = np.tile(zon.X0.center, (traj, 1))
        nz_ = vertex_noise(Wv, 0, traj, Tn)[:, :t]
        t0 = time.perf_counter(); out = co.simulate_batch(x0, nz_, A, B, threads=threads)
        d = time.perf_counter() - t0
        ok[0] = ok[0] and bool((out["status"] == 0).all())
        return d

    def rate(traj, threads):
        d_all = timed(traj, T, threads)
        d_warm = timed(traj, warmup, threads) if warmup > 0 else 0.0
        return traj * steps / max(d_all - d_warm, 1e-9)

    timed(max_thr, T, max_thr)                                                   # page in, spin the thread pool up
    d1 = timed(8, T, 1) / 8.0                                                    # seconds per trajectory on one thread (both legs ~ (1 + warmup / T) of it)
    per_traj = d1 * (1.0 + (warmup / T if warmup > 0 else 0.0))
    counts = sorted({c for c in (1, 2, 4, 8, 16, 32, 64, 128, 256, max_thr) if c <= max_thr})
    sample_s = min(2.0, 0.5 * seconds_budget / (len(counts) + 2 * repeats))          # ~2 s per point when the budget allows
    curve = []
    for c in counts:
        run_at = min(float(c), budget["effective_cpus"])                        # threads that can actually run at once
        traj = int(max(4 * c, min(32768, np.ceil(0.5 * sample_s * run_at / max(per_traj, 1e-9)))))
        curve.append({"threads": c, "value": float(max(rate(traj, c), rate(traj, c))), "trajectories": traj})    # best of two half-length samples (a shared host)
    # the best SUSTAINED point: the two highest points of the curve and the point at the CPU budget are sampled `repeats` times more and the highest median wins (a point
    # above the cgroup quota can burst for one sample and is throttled afterwards)
    finals = []
    cands = sorted(curve, key=lambda e: -e["value"])[:2]
    at_quota = min(curve, key=lambda e: abs(e["threads"] - budget["effective_cpus"]))     # the point at the CPU budget: bursts above it do not last
    if all(e is not at_quota for e in cands):
        cands.append(at_quota)
    for e in cands:
        sm = [float(rate(e["trajectories"], e["threads"])) for _ in range(max(repeats, 1))]
        finals.append((float(np.median(sm)), e, sm))
    med, best, samples = max(finals, key=lambda f: f[0])
    one = next(e["value"] for e in curve if e["threads"] == 1)
    for e in curve:
        e["speedup_over_one_thread"] = e["value"] / one
        e["parallel_efficiency"] = e["value"] / one / min(float(e["threads"]), budget["effective_cpus"])
    out = {"value": med, "unit": "MPC steps/s", "cores": best["threads"], "kind": "port",
           "effective_cpus": budget["effective_cpus"], "cpu_budget": budget, "scaling": curve,
           "samples": samples, "spread_rel": float((max(samples) - min(samples)) / med), "value_one_thread": float(one),
           "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_NUM_THREADS")},
           "reference_published": "~18 MPC steps/s: the reference's own pulley N=2 closed loop incl. build, 5 runs, hardware unstated "
                                  "(examples/results/pulley.tzddpc_times.npy; BASELINE.md) -- context only, the reference cannot run on this box"}
    if full_steps > 0:
        traj = best["trajectories"]
        x0 = np.tile(zon.X0.center, (traj, 1)); nz_ = vertex_noise(Wv, 0, traj, Tn)[:, :full_steps]
        fr = []
        for _ in range(2):
            t0 = time.perf_counter(); o = co.simulate_batch(x0, nz_, A, B, threads=best["threads"]); fr.append(traj * full_steps / (time.perf_counter() - t0))
            ok[0] = ok[0] and bool((o["status"] == 0).all())
        out["full_run"] = {"value": float(max(fr)), "unit": "MPC steps/s", "steps": full_steps, "threads": best["threads"], "trajectories": traj,
                           "samples": [float(v) for v in fr], "what": f"{full_steps} closed-loop steps from the centre of X0, nothing subtracted (best of 2)"}
    thr1 = _throttled()
    out["cgroup_throttled_periods_during_baseline"] = None if thr0 is None or thr1 is None else thr1 - thr0
    out["sample"] = (f"best point of a thread-scaling curve ({', '.join(str(c) for c in counts)} threads, ~{sample_s:.1f} s of work per point): {best['threads']} threads, "
                     f"median of {len(samples)} samples of {best['trajectories']} trajectories x closed-loop steps {warmup}..{T - 1} of the same {label} workload (time of {T} steps "
                     f"minus time of the first {warmup}), plain-C oracle with the same warm-started interior point, OpenMP over trajectories; the host offers "
                     f"{budget['usable_cpus']} of {budget['host_cpu_count']} logical CPUs (SMT {budget['smt_threads_per_core']}), cgroup quota "
                     f"{budget['cgroup_cpu_quota']} -> {budget['effective_cpus']:.0f} effective CPUs; {time.perf_counter() - t_wall0:.1f} s wall in total, all statuses zero: {ok[0]}")
    return out


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS) + sorted(GENSTACK_CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (default: the configuration's)")
    ap.add_argument("-ho", "--horizon", type=int, default=0, help="override the configuration's horizon (reference harness: -ho)")
    ap.add_argument("--repeats", type=int, default=0, help="timed windows (median reported; default 11)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-run-steps", type=int, default=50, help="T_sim of the untimed-nothing run from X0 (SURVEY 8d); 0 = skip")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo only with --dry-run")
    ap.add_argument("--dry-run", action="store_true", help="launch / shard / gather logic only (no GPU): CPU rehearsal of --gpus N")
    # flags of the reference's complexity harness (examples/1.double_integrator_computation_complexity.py:179-189)
    ap.add_argument("-m", "--method", default=None, help="TZDDPC (build_problem) | STZDDPC (build_problem_simplified(k0)); ZPC is another algorithm (out of scope)")
    ap.add_argument("-n", "--nevals", type=int, default=0, help="number of evaluations = timed windows")
    ap.add_argument("-k0", "--k0", type=int, default=None, help="k0 of the simplified problem (with -m STZDDPC)")
    args = ap.parse_args(argv)
    args.k0_override = "keep"
    if args.method is not None:
        meth = args.method.upper()
        if meth == "ZPC":
            ap.error("-m ZPC: the ZPC comparator (pyzpc) is a different algorithm and out of scope; use TZDDPC or STZDDPC")
        if meth not in ("TZDDPC", "STZDDPC"):
            ap.error(f"-m {args.method}: expected TZDDPC or STZDDPC")
        if args.config is None:
            args.config = "di_n20"                                   # the harness runs the double integrator (complexity-script zonotopes)
        if not args.horizon:
            args.horizon = 3                                         # the harness's default (-ho 3)
        args.k0_override = None if meth == "TZDDPC" else (1 if args.k0 is None else int(args.k0))
    elif args.k0 is not None:
        args.k0_override = int(args.k0)
    if args.config is None:
        args.config = "di_n20"
    if not args.repeats:
        args.repeats = args.nevals if args.nevals else 11
    if args.backend == "gloo" and not args.dry_run:
        ap.error("--backend gloo is the CPU rehearsal of the launch logic: add --dry-run")
    return args


def spawn_ranks(args, argv):
    """--gpus N > 1 outside torchrun: start the N ranks as fresh child processes (python -m torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1) BEFORE this process imports torch.cuda or makes any HIP call, relay their output, exit with their
    code.  Nothing is ever exec'ed from a process that has touched the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print(f"bench.py: the {args.gpus}-rank run failed (exit code {rc}): {' '.join(cmd)}", file=sys.stderr)
    return rc


def dry_run(args):
    """CPU rehearsal of the multi-rank path (gloo): sharding, per-rank noise by global trajectory index, the all-gather of
    (cost, final state), max-over-ranks timing -- no GPU, no solve: the per-trajectory 'result' is a checksum of its noise."""
    import torch
    import torch.distributed as dist
    from tzddpc_amd.dist import gather_results, shard_range, sync_calibration, vertex_noise
    from tzddpc_amd.harness import system
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    use_dist = "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    # the calibration exchange of the real run: TZ_DRYRUN_DIVERGE=1 makes every rank "choose" another push gain
    own_cal = [3, 0.1 * (1 + rank if os.environ.get("TZ_DRYRUN_DIVERGE") else 1), 0.01, 1e-5]
    cal, cal_same = sync_calibration(own_cal)
    sysname = CONFIGS[args.config][0]
    Bl = args.batch or CONFIGS[args.config][5]
    A, Bm, zon, _ = system(sysname)
    n = A.shape[0]
    total = Bl * world
    lo, hi = shard_range(total, world, rank)
    Wv = zon.W.compute_vertices()
    K = args.steps

    def results(first, count):
        nz_ = vertex_noise(Wv, first, count, K)
        return torch.from_numpy(np.concatenate([nz_.sum(axis=(1, 2))[:, None], nz_.sum(axis=1)], axis=1))
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    local = results(lo, hi - lo)
    gathered = gather_results(local, total)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if use_dist:
        tl = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tl, torch.tensor([elapsed], dtype=torch.float64))
        per_rank = [float(t.item()) for t in tl]
    if rank == 0:
        full = results(0, total)
        print(json.dumps({"metric": "dry run (launch / shard / gather logic only)", "value": None, "unit": "MPC steps/s", "n_gpus": world, "steps": K,
                          "warmup": args.warmup, "dry_run": True, "backend": args.backend, "scaling": "weak",
                          "world_size_read_back": dist.get_world_size() if use_dist else 1,
                          "config": {"name": args.config, "trajectories_per_gpu": Bl, "gathered_rows": int(gathered.shape[0])},
                          "calibration_identical_across_ranks": bool(cal_same), "calibration_adopted_from_rank0": cal,
                          "gather_matches_unsharded": bool(torch.equal(gathered, full)), "rank_window_ms": [t * 1e3 for t in per_rank]}))
    if use_dist:
        dist.destroy_process_group()
    return 0


def bench_genstack(args, local_rank=0):
    """K1g (tz_genstack_kernel): one "step" = the interval hulls of all N literal tubes for every trajectory of the
    batch (one pass over the stacked generator map in HBM per tile of 256 trajectories).  The binding roofline depends on the batch:
    the stack is streamed once per 256-trajectory tile, so few trajectories are HBM-bound and many are f64-FMA-bound."""
    import torch
    from tzddpc_amd import TZDDPC, native
    from tzddpc_amd.genstack import build_stack, evaluate_host
    from tzddpc_amd.harness import generate_trajectories, system
    sysname, N, k0, Bdef, desc = GENSTACK_CONFIGS[args.config]
    N = args.horizon or N
    if args.k0_override != "keep":
        k0 = args.k0_override
    B = args.batch or Bdef
    A, Bm, zon, T = system(sysname)
    n, m = Bm.shape
    ctl = TZDDPC(generate_trajectories(A, Bm, zon.X0, zon.U, zon.W, 1, T, np.random.default_rng(25)), device=local_rank)
    ctl.build_zonotopes_theta(zon)
    t0 = time.time(); st = build_stack(ctl.MdataK, ctl.Mdelta, ctl.theta.K, zon.W, n, m, N, k0, nseg=N); t_build = time.time() - t0
    gs = native.GenStack(local_rank, st)
    info = gs.info()
    rng = np.random.default_rng(3)
    e0 = 0.02 * rng.standard_normal((B, n)); zeta = rng.standard_normal((B, N, n + m))
    dev = torch.device("cuda", local_rank)
    te, tz = torch.from_numpy(e0).to(dev), torch.from_numpy(zeta).to(dev)
    c = torch.empty((B, N, n), dtype=torch.float64, device=dev); rx = torch.empty_like(c); ru = torch.empty((B, N, m), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    call = lambda: gs.intervals_ptr(B, te.data_ptr(), tz.data_ptr(), c.data_ptr(), rx.data_ptr(), ru.data_ptr())
    for _ in range(max(args.warmup, 1)):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms = [call() for _ in range(args.steps)]                  # HIP events inside the library around the streaming kernel
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ref = evaluate_host(st, e0[B // 2], zeta[B // 2])
    err = max(np.abs(rx[B // 2].cpu().numpy() - np.array([r[1] for r in ref])).max() / (1 + max(r[1].max() for r in ref)),
              np.abs(c[B // 2].cpu().numpy() - np.array([r[0] for r in ref])).max())
    med = float(np.median(ms))
    tiles = (B + 255) // 256
    bytes_alg = info["stack_bytes"] * tiles + B * (n + N * (n + m)) * 8 + info["chunks"] * B * (n + m) * 8
    flop = info["generators"] * B * (2 * n * (n + m) + 2 * n + 2 * m * n + n + m)
    gbs, tfs = bytes_alg / (med * 1e-3) / 1e9, flop / (med * 1e-3) / 1e12
    hbm_bound = gbs / 8000.0 > tfs / F64_MFMA_PEAK_TFLOPS
    roof = ({"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0} if hbm_bound else
            {"bound": "mfma", "achieved": tfs, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfs / F64_MFMA_PEAK_TFLOPS})
    traffic = tsrc = None
    try:                                                      # HBM-side bytes of one launch from the PMC passes archived under profiles/
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))["configs"][f"{args.config}_b{B}"]
        if N == GENSTACK_CONFIGS[args.config][1] and k0 == GENSTACK_CONFIGS[args.config][2]:
            traffic, tsrc = float(e["fetch_bytes_per_launch"]) + float(e["write_bytes_per_launch"]), e["source"]
    except Exception:
        pass
    roof.update({"traffic": traffic, "traffic_source": tsrc, "kernel": "tz_genstack_mfma_kernel" if 3 <= n + m <= 7 else "tz_genstack_kernel", "avg_launch_ms": med, "launches": len(ms),
                 "algorithmic_bytes_per_launch": bytes_alg, "algorithmic_flop_per_launch": flop,
                 "other_roof": {"hbm_frac": gbs / 8000.0, "f64_frac": tfs / F64_MFMA_PEAK_TFLOPS},
                 "note": "algorithmic bytes: the stack once per tile of 256 trajectories at 8 n (1 + n + m) bytes per generator (the kernel's own layout carries "
                         "the m rows K M as well: (n + m)(1 + n + m) doubles per generator, +20 % for n = 5, m = 1); algorithmic flop: 2 n (n + m) + 2 n + 2 m n + n + m "
                         "f64 flop per generator and trajectory against the 78.6 TFLOP/s f64 matrix (= vector) peak; the larger of the two fractions binds"})
    print(json.dumps({"metric": f"literal tube evaluations/s (K1g), {args.config}", "value": B * args.steps / elapsed, "unit": "trajectory tube-stack evaluations/s",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": f"{desc}; {sysname} n={n} m={m}, N={N}, k0={k0}, {info['generators']} generators ({info['stack_bytes'] / 1e6:.1f} MB), {B} trajectories",
                                 "name": args.config, "generators": info["generators"], "max_generators_per_tube": int(st.num_generators.max()),
                                 "chunks": info["chunks"], "trajectories": B, "stack_build_s": t_build},
                      "roofline": roof, "max_err_vs_numpy": float(err)}))
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    in_torchrun = "RANK" in os.environ
    if args.gpus > 1 and not in_torchrun:
        return spawn_ranks(args, argv)                      # first thing: nothing has touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if in_torchrun and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        return 2
    # OMP_PROC_BIND / OMP_PLACES are left as the caller set them (reported in cpu_baseline.omp): on the GPU box OMP_PLACES=cores cut the
    # OpenMP team of the baseline to 2 threads of the 128 the container may use
    if args.dry_run:
        return dry_run(args)

    ensure_built()
    import torch
    import torch.distributed as dist
    from tzddpc_amd.dist import gather_results, shard_range, sync_calibration, vertex_noise

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.config in GENSTACK_CONFIGS:
        if world != 1:
            print("bench.py: the K1g configurations are single-GPU lines", file=sys.stderr)
            return 2
        return bench_genstack(args, local_rank)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or in_torchrun                     # under torch.distributed.run even one rank goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
    dev = torch.device("cuda", local_rank)

    t_build0 = time.perf_counter()
    ctl, A, Bm, zon, horizon, k0 = build_controller(local_rank, args.config, args.horizon, args.k0_override)
    build_s = time.perf_counter() - t_build0
    nat = ctl._native
    # every rank calibrated on its own device; all of them run with rank 0's choice (the window is the max over ranks: a rank with
    # another push gain / stopping target would silently set the job's time), and the line says whether they had agreed anyway
    own_cal = [int(ctl.warm_shift_policy), float(ctl.warm_push_gain), float(ctl.warm_push_cap), float(ctl.mu_factor)]
    cal, cal_same = sync_calibration(own_cal, device=dev if use_dist else None)
    if cal != own_cal:
        nat.set_warm_shift(int(cal[0])); nat.set_warm_push(1e-8, cal[1], cal[2]); nat.set_stopping(100.0, cal[3])
        ctl.warm_shift_policy, ctl.warm_push_gain, ctl.warm_push_cap, ctl.mu_factor = int(cal[0]), cal[1], cal[2], cal[3]
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
    Tfull = max(0, args.full_run_steps)
    Wv = zon.W.compute_vertices()
    nsteps = max(K + W, Tfull)
    noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(Wv, lo, Bl, nsteps).transpose(1, 0, 2))).to(dev)   # steps x B x n
    x_same = torch.from_numpy(np.tile(zon.X0.center, (Bl, 1))).to(dev)
    # "x0 = X0 center for all (optionally + box jitter +-0.25, seed 7)": drawn for the whole job, sliced per rank
    jit = np.random.default_rng(7).uniform(-0.25, 0.25, size=(total, n))[lo:hi]
    x_jit = torch.from_numpy(np.tile(zon.X0.center, (Bl, 1)) + jit).to(dev)
    x = x_same.clone(); xbar = x.clone(); e = torch.zeros_like(x)
    u = torch.zeros((Bl, m), dtype=torch.float64, device=dev)
    cost = torch.zeros(Bl, dtype=torch.float64, device=dev)
    status = torch.zeros(Bl, dtype=torch.int32, device=dev)
    At = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev)
    Bt = torch.from_numpy(np.ascontiguousarray(Bm, dtype=np.float64).reshape(n, m)).to(dev)
    torch.cuda.synchronize()

    ptrs = (x.data_ptr(), xbar.data_ptr(), e.data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), status.data_ptr())
    res = torch.empty((Bl, 1 + n), dtype=torch.float64, device=dev)          # per-trajectory [cost | final state]: what the ranks exchange

    def run(first_step, k):   # k closed-loop steps, issued by one C call (tz_mpc_run): no host work between steps
        nat.mpc_run_ptr(Bl, k, ptrs[0], ptrs[1], ptrs[2], noise[first_step].data_ptr(), ptrs[3], ptrs[4], ptrs[5], ptrs[6], ptrs[7])

    def collect():
        torch.cat((cost.unsqueeze(1), x), dim=1, out=res)                    # one copy kernel inside the timed region
        return gather_results(res, total)

    def fresh_start(x_start, warm):   # state back to the start, no memory of earlier solves in the handle, `warm` untimed warm-up steps
        x.copy_(x_start); xbar.copy_(x_start); e.zero_()
        torch.cuda.synchronize()
        nat.reset_warm()
        if warm > 0:
            run(0, warm)
        nat.sync()
        return (status != 0).int()

    def measure(x_start, warm, k, reps, discard=0):
        """`reps` timed windows of k steps, each after a fresh start + `warm` untimed steps; barrier + synchronize on both sides,
        max over ranks.  `discard` windows of the same kind run first IN THE SAME LOOP and are dropped."""
        out = dict(windows=[], kern_ms=[], facts=[], solves=[], fmax=[], rank_ms=[], gathered=None, bad=torch.zeros(Bl, dtype=torch.int32, device=dev))
        for rep in range(discard + reps):
            if rep == discard:
                for key in ("windows", "kern_ms", "facts", "solves", "fmax", "rank_ms"):
                    out[key] = []
            out["bad"] |= fresh_start(x_start, warm)
            nat.timing_enable(True)                                          # zeroes the event sums / work counters of the library
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize(); nat.sync()
            t0 = time.perf_counter()
            run(warm, k)
            out["gathered"] = collect()                                      # per-trajectory cost + final state only
            torch.cuda.synchronize()
            mine = time.perf_counter() - t0                                  # this rank's K steps + gather, device idle again
            if use_dist:
                dist.barrier()                                               # the closing barrier of the bracket: no rank starts the next window early
            # the job's time is the MAX over ranks of their own bracketed times (the clock is read before the closing barrier: its own
            # latency -- an RCCL launch, 60 us against a 0.7 ms window -- is not part of the K steps)
            tmax = torch.tensor([mine], dtype=torch.float64, device=dev)
            if use_dist:
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                tl = torch.zeros(world, dtype=torch.float64, device=dev)
                dist.all_gather_into_tensor(tl, torch.tensor([mine], dtype=torch.float64, device=dev))
                out["rank_ms"].append([float(v) * 1e3 for v in tl.cpu()])
            else:
                out["rank_ms"].append([mine * 1e3])
            out["windows"].append(float(tmax.item()))
            ms, cnt = nat.timing_get(1)                                      # HIP events recorded by the library on its own stream around the launch
            work = nat.work_get()
            out["kern_ms"].append(ms / max(cnt, 1)); out["facts"].append(work["factorizations"]); out["solves"].append(work["trajectory_solves"]); out["fmax"].append(work["max_factorizations_one_trajectory"])
            out["bad"] |= (status != 0).int()
        nb = out["bad"].sum().to(torch.float64).reshape(1)
        if use_dist:
            dist.all_reduce(nb, op=dist.ReduceOp.SUM)
        out["nbad"] = int(nb.item())
        order = np.argsort(out["windows"])
        out["med"] = int(order[len(order) // 2])                             # the median window and ITS work counters
        return out

    def summary(mm, k):
        i = mm["med"]
        return {"value": total * k / mm["windows"][i], "unit": "MPC steps/s", "steps": k, "window_ms": mm["windows"][i] * 1e3,
                "window_ms_all": [round(w * 1e3, 4) for w in mm["windows"]], "kernel_ms": mm["kern_ms"][i],
                "ipm_factorizations_per_trajectory_step": mm["facts"][i] / max(mm["solves"][i], 1),
                "ipm_factorizations_slowest_trajectory": int(mm["fmax"][i]),      # all trajectories are resident at once: the launch lasts as long as this one
                "unsolved_trajectory_steps": mm["nbad"]}

    fresh_start(x_same, W)
    _ = collect()                                                        # warm torch's copy / RCCL paths outside the timed region
    DISCARD = 30                                                         # discarded windows: the device reaches its sustained state only after ~13
                                                                         # windows of this size (0.63 -> 0.585 ms kernel time, profiles/r4_window_trend.txt)
    head = measure(x_same, W, K, R, DISCARD)                             # (clocks, instruction / constant caches); then the driver-contract window:
                                                                         # W untimed steps from X0, then K timed, R times from a fresh start
    extra = {}
    if Tfull > 0:
        # SURVEY 8d's metric as written: B * T_sim / wall, T_sim = 50 from X0 with NOTHING untimed (the reference times build + all
        # 200 steps, examples/2.pulley_sim.py:79-97); the transient's 9-13 iteration steps are inside
        extra["full_run"] = summary(measure(x_same, 0, Tfull, 5, 10), Tfull)
        extra["full_run"]["what"] = f"{Tfull} closed-loop steps from the centre of X0, no untimed warm-up, one launch; median of 5"
        jw = measure(x_jit, W, K, 5, 10)
        extra["jittered_start"] = summary(jw, K)
        extra["jittered_start"]["what"] = (f"every trajectory from its own point of X0 + U(-0.25, 0.25)^n (seed 7, SURVEY 8d config 2), {W} untimed + {K} timed "
                                           "steps as the headline; median of 5")
        extra["jittered_start"]["full_run"] = summary(measure(x_jit, 0, Tfull, 5, 10), Tfull)

    if rank == 0:
        windows, kern_ms, facts, solves, med = head["windows"], head["kern_ms"], head["facts"], head["solves"], head["med"]
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
        sysname, _, _, _, _, _, desc = CONFIGS[args.config]
        label = f"{args.config} (N={horizon}" + (f", k0={k0})" if k0 is not None else ")")
        headline = args.config == "di_n20" and horizon == 20 and k0 is None
        line = {
            "metric": "MPC steps/sec (batched trajectories), double-integrator N=20" if headline
                      else f"MPC steps/sec (batched trajectories), {label}",
            "value": total * K / elapsed, "unit": "MPC steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{desc}; {Bl} closed-loop trajectories per GPU, {'complexity-script' if sysname in ('di_cc', 'di2in') else 'example'} zonotopes, "
                                   f"vertex-of-W noise PCG64(1000+i), all {K} timed steps in one launch",
                       "name": args.config, "trajectories_per_gpu": Bl, "horizon": horizon, "k0": k0, "nz": ctl.qp.nz, "rows": int(nat.mi),
                       "ipm_factorizations_per_trajectory_step": iters_mean, "ipm_factorizations_slowest_trajectory": int(head["fmax"][med]), "warm_start": True,
                       "stored_start": (None if getattr(ctl, "stored_start", None) is None else [float(v) for v in ctl.stored_start[0]]),
                       "warm_shift_policy": int(ctl.warm_shift_policy), "warm_push_gain": float(ctl.warm_push_gain), "warm_push_cap": (None if not np.isfinite(ctl.warm_push_cap) else float(ctl.warm_push_cap)), "mu_factor": float(ctl.mu_factor), "unsolved_trajectory_steps": head["nbad"],
                       "gathered_rows": int(head["gathered"].shape[0]), "world_size_read_back": (dist.get_world_size() if use_dist else 1),
                       "lds_bytes_per_workgroup": nat.plan_info()["lds_bytes"],
                       "build_seconds": build_s, "calibration_seconds": float(getattr(ctl, "calibration_seconds", 0.0)),
                       "calibrated_at_build": getattr(ctl, "calibrated", None),
                       "calibration_identical_across_ranks": bool(cal_same), "calibration_adopted_from_rank0": cal,
                       "env_overrides": {k: v for k, v in os.environ.items() if k.startswith("TZ_")}},
            "timing": {"repeats": R, "reported": "median window", "window_ms": [round(w * 1e3, 4) for w in windows],
                       "window_ms_min": min(windows) * 1e3, "window_ms_max": max(windows) * 1e3,
                       "spread_rel": (max(windows) - min(windows)) / elapsed,
                       "value_best": total * K / min(windows), "value_worst": total * K / max(windows),
                       "rank_window_ms_of_reported": head["rank_ms"][med],
                       "discarded_windows_before": DISCARD,
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
        line.update(extra)
        tb, tinfo = archived_traffic(args.config, ctl.qp.nz, nat.mi, Bl, K)
        if tb is not None:
            line["roofline"]["traffic"] = tb
            line["roofline"]["traffic_source"] = tinfo
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cb = cpu_baseline(ctl, A, Bm, zon, label, W, K, full_steps=Tfull)
                if cb.get("value"):                       # ratios against the BEST point of the CPU curve; never a quality claim (the roofline is)
                    cb["gpu_over_cpu"] = line["value"] / cb["value"]
                    cb["gpu_over_one_thread_x_effective_cpus"] = line["value"] / (cb["value_one_thread"] * cb["effective_cpus"])
                if cb.get("full_run") and "full_run" in line:
                    cb["full_run"]["gpu_over_cpu"] = line["full_run"]["value"] / cb["full_run"]["value"]
            except Exception as ex:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "MPC steps/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
