#!/bin/bash
# A/B of library builds on ONE box (box-to-box spread of the kernel time is 1-2 %, run-to-run on a box 0.2 %):
#   tools/ab.sh tzddpc_amd/lib/ab/base.so tzddpc_amd/lib/libtzddpc_hip.so ...   -> kernel ms of the timed launch and steps/s, three runs each
for lib in "$@"; do
  for i in 1 2 3; do
    TZ_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --full-run-steps 0 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(l['roofline']['avg_launch_ms'],4), int(l['value']), l['config']['ipm_factorizations_per_trajectory_step'], l['config']['unsolved_trajectory_steps'])" || exit 1
  done
done
