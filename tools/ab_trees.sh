#!/bin/bash
# tools/ab_trees.sh <other-tree-dir> cfg...: the bench line of every configuration from this tree and from another checkout staged inside
# the repository (its own python + library), alternating, on one box: value, kernel ms, factorisations per step
other=$1; shift
line() { python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', round(l['value']/1e6,3), 'M  kernel', round(l['roofline']['avg_launch_ms'],4), 'ms  fact/step', round(l['config']['ipm_factorizations_per_trajectory_step'],3), ' push', l['config']['warm_push_gain'], l['config']['warm_push_cap'], 'mu', l['config']['mu_factor'], 'shift', l['config']['warm_shift_policy'])"; }
for cfg in "$@"; do
  for rep in 1 2; do
    (cd $other && timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --full-run-steps 0 2>/dev/null | line other $cfg) || echo "other $cfg FAILED"
    timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --full-run-steps 0 2>/dev/null | line this $cfg || echo "this $cfg FAILED"
  done
done
