"""Turn the raw output of tools/profile_round.sh <tag> <config> (under gpurun_out/) into the files committed under profiles/:
<tag>_<config>_bench.json (the driver-style line: 11 windows, cpu baseline), _kernel_stats.csv, _pmc.json, _summary.json, and the
entry of that configuration in traffic_latest.json (HBM-side bytes per trajectory-step, read and scaled by bench.py).
python tools/summarize_round.py r2a di_n20"""
import csv, json, os, shutil, sys
sys.path.insert(0, ".")
tag, cfg = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "di_n20")
t = f"{tag}_{cfg}"
src, dst = "gpurun_out", "profiles"
bench = json.loads(open(f"{src}/{t}_bench.json").read().strip().splitlines()[-1])
pmc = json.load(open(f"{src}/{t}_pmc.json"))
kname = next(k for k in pmc if "tz_ipm_kernel" in k)
launches = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
            for r in csv.DictReader(open(f"{src}/{t}_trace/t_kernel_trace.csv")) if "tz_ipm_kernel" in r["Kernel_Name"]]
launch_ms = [d for _, d in sorted(launches)]
c = pmc[kname]
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B,
# i.e. half the bytes of wide streaming reads -- doubled here (an upper bound for narrow reads; other widths are uncalibrated); WRITE_SIZE is exact.
fetch = 2.0 * c["FETCH_SIZE"][-1] * 1024.0
write = c["WRITE_SIZE"][-1] * 1024.0
Bl, K = bench["config"]["trajectories_per_gpu"], bench["steps"]
import __graft_entry__
entry = {"nz": bench["config"]["nz"], "rows": bench["config"]["rows"], "steps": K, "trajectories_per_gpu": Bl,
         "fetch_bytes_per_trajectory_step": fetch / (Bl * K), "write_bytes_per_trajectory_step": write / (Bl * K),
         "lib_source_hash": __graft_entry__.source_hash(),
         "source": f"profiles/{t}_pmc.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, timed launch of `python3 bench.py --config {cfg} "
                   f"--steps {K} --repeats 1 --no-cpu-baseline`; KiB -> bytes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)"}
try:
    tr = json.load(open(f"{dst}/traffic_latest.json"))
    assert "configs" in tr
except Exception:
    tr = {"configs": {}}
tr["configs"][cfg] = entry
waves = Bl * 4 * K
summary = {
    "kernel": kname,
    "launch_ms_rocprof_all_(calibration_launches,_warm-up,_timed)": launch_ms,
    "timed_launch_ms_rocprof": launch_ms[-1],
    "timed_launch_ms_hip_events_median_of_bench_line": bench["roofline"]["avg_launch_ms"],
    "steps_in_timed_launch": K, "trajectories": Bl,
    "value_steps_per_s": bench["value"], "timing": bench.get("timing"), "cpu_baseline": bench.get("cpu_baseline"), "roofline": bench["roofline"], "config": bench["config"],
    "hbm_side_bytes_timed_launch": {"fetch": fetch, "write": write, "per_trajectory_step": (fetch + write) / (Bl * K)},
    "per_wave_per_step_timed_launch": {k: round(v[-1] / waves, 1) for k, v in c.items() if k.startswith("SQ_INSTS") or k.startswith("SQ_LDS")},
    "wave_cycle_fractions_timed_launch": {k: round(c[k][-1] / c["SQ_WAVE_CYCLES"][-1], 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in c and "SQ_WAVE_CYCLES" in c},
    "pmc_per_launch_in_dispatch_order": c,
}
json.dump(summary, open(f"{dst}/{t}_summary.json", "w"), indent=1)
json.dump(tr, open(f"{dst}/traffic_latest.json", "w"), indent=1)
json.dump(bench, open(f"{dst}/{t}_bench.json", "w"), indent=1)
json.dump(pmc, open(f"{dst}/{t}_pmc.json", "w"), indent=1)
shutil.copy(f"{src}/{t}_kernel_stats.csv", f"{dst}/{t}_kernel_stats.csv")
print(json.dumps({k: summary[k] for k in ("timed_launch_ms_rocprof", "timed_launch_ms_hip_events_median_of_bench_line", "value_steps_per_s", "hbm_side_bytes_timed_launch",
                                          "per_wave_per_step_timed_launch", "wave_cycle_fractions_timed_launch")}, indent=1))
