#!/bin/bash
# tools/devlib.sh <name> [-Dflags...]  -> tzddpc_amd/lib/ab/<name>.so with only the bench-problem kernel variant (seconds), prints its resources
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tzddpc_amd/lib/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -disable-machine-licm -shared -fPIC -DTZ_ONLY_SMALL "$@" -Rpass-analysis=kernel-resource-usage tzddpc_amd/csrc/tzddpc_hip.hip -Iinclude -o tzddpc_amd/lib/ab/$name.so 2> /tmp/devlib_$name.txt || { tail -20 /tmp/devlib_$name.txt; exit 1; }
grep -A12 "tz_ipm_kernelILi1ELi1ELi4" /tmp/devlib_$name.txt | grep -i "scratch\|VGPRs:\|SGPRs:" | sed 's/.*remark: *//' | tr '\n' ' '; echo
