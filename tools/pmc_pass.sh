#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]   (one rocprofv3 --pmc pass of bench.py; prints per-launch means for tz_ipm_kernel)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" -d gpurun_out/pmc_$tag -o p --output-format csv -- python3 bench.py --no-cpu-baseline --steps 20 > gpurun_out/pmc_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, collections, sys
f = glob.glob(f"gpurun_out/pmc_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    if "tz_ipm_kernel" not in row["Kernel_Name"]: continue
    acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
print({c: round(v / n[c]) for c, v in acc.items()})
PY
