#!/bin/bash
# on the GPU box: correctness + speed of the tile-triangle dev variants (tools/devlib_tt.sh tt322 3,2,2; tt322p ... -DTZ_PROFILE=1; tt531 5,3,1)
out=gpurun_out/$1; shift
A=tzddpc_amd/lib/ab
{
TZ_LIB=$A/tt322.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "horizon_sweep_against_c_oracle[di_n40] or closed_loop_configs_against_c_oracle[dim5_n20 or golden_parity[dim5_n20] or tube_theta_matches_oracle[dim5_n20]" 2>&1 | tail -5
[ -n "$SKIP80" ] || TZ_LIB=$A/tt531.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "horizon_sweep_against_c_oracle[di_n80]" 2>&1 | tail -5
for c in di_n40 dim5_n20; do TZ_LIB=$A/tt322p.so timeout -k 10 200 python tools/gpu_prof_multi.py $c 10 2>&1 | grep -v amdgpu.ids; done
for c in di_n40 dim5_n20; do TZ_LIB=$A/tt322.so timeout -k 10 120 python tools/gpu_throughput.py $c 1024 2>&1 | grep -v amdgpu.ids; done
[ -n "$SKIP80" ] || TZ_LIB=$A/tt531.so timeout -k 10 120 python tools/gpu_throughput.py di_n80 1024 2>&1 | grep -v amdgpu.ids
} > $out 2>&1
cat $out
