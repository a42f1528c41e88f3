#!/bin/bash
for lib in tzddpc_amd/lib/ab/g4.so tzddpc_amd/lib/ab/g16d4.so; do
for b in 128 256 512 1024 2048; do
  TZ_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --full-run-steps 0 --batch $b 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib batch $b', round(l['roofline']['avg_launch_ms'],4), int(l['value']), l['config']['ipm_factorizations_per_trajectory_step'])"
done; done
