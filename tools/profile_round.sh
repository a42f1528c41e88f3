#!/bin/bash
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh <tag> [config] [steps]
# The driver-style bench line (11 windows, cpu baseline), then one rocprofv3 --kernel-trace --stats run and separate --pmc passes of
# `bench.py --config <config> --steps <steps> --repeats 1 --no-cpu-baseline` (the LAST tz_ipm_kernel launch of such a run is the
# timed one); raw output under gpurun_out/<tag>_<config>_*, summarised by tools/summarize_round.py <tag> <config> into profiles/.
tag=${1:-rXX}; cfg=${2:-di_n20}; steps=${3:-20}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out; t=${tag}_${cfg}
python3 bench.py --config $cfg --steps $steps --warmup 5 > $out/${t}_bench.json 2> $out/${t}_bench.err || exit 1
B="python3 bench.py --config $cfg --steps $steps --warmup 5 --repeats 1 --no-cpu-baseline --full-run-steps 0"   # no full-run / jitter legs: the timed window is the LAST launch
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${t}_trace -o t -- $B > $out/${t}_trace.log 2>&1 || exit 1
cp $(find $out/${t}_trace -name "*kernel_stats.csv" | head -1) $out/${t}_kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $out/${t}_pmc_$name -o p -- $B > $out/${t}_pmc_$name.log 2>&1 || exit 1
done
python3 - "$t" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]; res = {}
for f in glob.glob(f"gpurun_out/{tag}_pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # (kernel, dispatch) -> counter -> value
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "tz_" not in k: continue
        per[(k, int(row["Dispatch_Id"]))][row["Counter_Name"]] += float(row["Counter_Value"])
    for (k, d) in sorted(per):
        for c, v in per[(k, d)].items():
            res.setdefault(k, {}).setdefault(c, []).append(v)       # launches in dispatch order: [calibration..., warm-up launch, timed launch]
json.dump(res, open(f"gpurun_out/{tag}_pmc.json", "w"), indent=1)
print({k: {c: v[-1] for c, v in d.items()} for k, d in res.items() if "ipm" in k})
PY
