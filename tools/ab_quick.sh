#!/bin/bash
# tools/ab_quick.sh lib1.so lib2.so ...: per library the driver-window line (kernel ms, steps/s, factorisations per step, unsolved) three times,
# then the steady state (tools/gpu_throughput.py di_n20 1024 30 10) -- one box, back to back
for lib in "$@"; do
  for i in 1 2 3; do
    TZ_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --full-run-steps 0 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(l['roofline']['avg_launch_ms'],4), int(l['value']), l['config']['ipm_factorizations_per_trajectory_step'], l['config']['unsolved_trajectory_steps'])" || exit 1
  done
  TZ_LIB=$lib timeout -k 10 300 python tools/gpu_throughput.py di_n20 1024 30 10 2>/dev/null | cut -c1-160
  TZ_LIB=$lib timeout -k 10 300 python tools/gpu_throughput.py di_n20 1024 50 0 2>/dev/null | cut -c1-160
done
