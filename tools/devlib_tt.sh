#!/bin/bash
# tools/devlib_tt.sh <name> <R,NCG,W> [-Dflags...]  -> tzddpc_amd/lib/ab/<name>.so with ONE tile-triangle kernel variant (seconds), prints its resources
cd "$(dirname "$0")/.."
name=$1; var=$2; shift; shift
mkdir -p tzddpc_amd/lib/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -disable-machine-licm -shared -fPIC "-DTZ_ONLY_TT=$var" "$@" -Rpass-analysis=kernel-resource-usage tzddpc_amd/csrc/tzddpc_hip.hip -Iinclude -o tzddpc_amd/lib/ab/$name.so 2> /tmp/devlib_$name.txt || { grep -E "error" -A5 /tmp/devlib_$name.txt | head -40; exit 1; }
grep -A12 "tz_ipm_kernel" /tmp/devlib_$name.txt | grep -i "scratch\|VGPRs:\|SGPRs:\|Spill" | sed 's/.*remark: *//' | tr '\n' ' '; echo
