#!/bin/bash
# on the GPU box: sensitivity of the factorisation count to the warm-start constants (env overrides of tz_problem_create)
out=gpurun_out/$1; shift
{
for c in "$@"; do
  for g in 3 1 0.5 0.3 0.1 0.03 0.01 0.003 0; do
    echo -n "gain=$g  "; TZ_WARM_GAIN=$g timeout -k 10 120 python tools/gpu_throughput.py $c 1024 2>&1 | grep -v amdgpu.ids | cut -c1-175
  done
done
} > $out 2>&1
cat $out
