#!/bin/bash
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r1c
# One rocprofv3 --kernel-trace --stats run and separate --pmc passes of bench.py; summaries land in gpurun_out/<tag>_*.
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o t -- python3 bench.py --no-cpu-baseline > $out/${tag}_trace.log 2>&1 || exit 1
cp $(find $out/${tag}_trace -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $out/${tag}_pmc_$name -o p -- python3 bench.py --no-cpu-baseline > $out/${tag}_pmc_$name.log 2>&1 || exit 1
done
python3 - "$tag" <<'PY'
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
            res.setdefault(k, {}).setdefault(c, []).append(v)       # launches in dispatch order: [warm-up launch, timed launch]
json.dump(res, open(f"gpurun_out/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
