#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_aux.sh <tag>
# rocprofv3 --kernel-trace --stats of the auxiliary kernels (K0, spectral radius, adversary) and of K1g at 1024 and 32 trajectories, plus the
# FETCH_SIZE / WRITE_SIZE passes of K1g (separate --pmc runs); raw output under gpurun_out/<tag>_aux_*, copied to profiles/ by hand.
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_aux_trace -o t -- python3 tools/gpu_aux_kernels.py 5 > $out/${tag}_aux.log 2>&1 || exit 1
cp $(find $out/${tag}_aux_trace -name "*kernel_stats.csv" | head -1) $out/${tag}_aux_kernel_stats.csv
for b in 1024 32; do
  B="python3 bench.py --config genstack_dim5_k1 --steps 20 --warmup 3 --batch $b"
  $B > $out/${tag}_genstack_b${b}_bench.json 2> /dev/null || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_genstack_b${b}_trace -o t -- $B > $out/${tag}_genstack_b${b}_trace.log 2>&1 || exit 1
  cp $(find $out/${tag}_genstack_b${b}_trace -name "*kernel_stats.csv" | head -1) $out/${tag}_genstack_b${b}_kernel_stats.csv
  for pass in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --output-format csv -d $out/${tag}_genstack_b${b}_pmc_$name -o p -- $B > $out/${tag}_genstack_b${b}_pmc_$name.log 2>&1 || exit 1
  done
done
python3 - "$tag" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]; res = {}
for b in (1024, 32):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{tag}_genstack_b{b}_pmc_*/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            if "tz_genstack" not in row["Kernel_Name"]: continue
            acc[(row["Kernel_Name"].split("(")[0][:60], int(row["Dispatch_Id"]))][row["Counter_Name"]] += float(row["Counter_Value"])
        for (k, d) in sorted(acc):
            for c, v in acc[(k, d)].items(): per[k][c].append(v)
    res[f"b{b}"] = {k: {c: {"per_launch_median": sorted(v)[len(v) // 2], "launches": len(v)} for c, v in d.items()} for k, d in per.items()}
json.dump(res, open(f"gpurun_out/{tag}_genstack_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
