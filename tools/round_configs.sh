#!/bin/bash
# usage (GPU box, repo root): bash tools/round_configs.sh <tag>  -- the driver-style bench line of every configuration -> gpurun_out/<tag>_cfg_<config>_bench.json
tag=${1:-rXX}
for cfg in di_n20 di_n5 di_n10 di_n40 di_n80 di_n20_k1 di_n20_k2 pulley_n10 dim5_n20 dim5m2_n20; do
  python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_cfg_${cfg}_bench.json 2> gpurun_out/${tag}_cfg_${cfg}_bench.err || echo "FAILED $cfg"
  python3 - "$tag" "$cfg" <<'PY'
import json, sys
l = json.loads(open(f"gpurun_out/{sys.argv[1]}_cfg_{sys.argv[2]}_bench.json").read().strip().splitlines()[-1])
c = l["config"]
print(f"{sys.argv[2]:12s} nz {c['nz']:4d} rows {c['rows']:5d}  window {l['value'] / 1e6:7.3f} M  fact/step {c['ipm_factorizations_per_trajectory_step']:.2f}  full_run {l['full_run']['value'] / 1e6:7.3f} M ({l['full_run']['ipm_factorizations_per_trajectory_step']:.2f})  "
      f"jitter {l['jittered_start']['value'] / 1e6:7.3f} M  frac {l['roofline']['frac']:.3f}  mu {c['mu_factor']} push {c['warm_push_gain']},{c['warm_push_cap']} shift {c['warm_shift_policy']}  unsolved {c['unsolved_trajectory_steps']}")
PY
done
