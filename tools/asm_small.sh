#!/bin/bash
# Device assembly of the bench-problem variant of tz_ipm_kernel only (-DTZ_ONLY_SMALL: ~20 s instead of minutes) -> /tmp/di.s, with the
# register / scratch summary.  tools/asm_small.sh lib <out.so> builds a library with only that variant (for tools/ab.sh).
cd "$(dirname "$0")/.."
if [ "$1" = "lib" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -disable-machine-licm -shared -fPIC -DTZ_ONLY_SMALL tzddpc_amd/csrc/tzddpc_hip.hip -Iinclude -o "$2"
  exit $?
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -disable-machine-licm --cuda-device-only -S -DTZ_ONLY_SMALL tzddpc_amd/csrc/tzddpc_hip.hip -Iinclude -o /tmp/k_small.s || exit 1
awk '/^_Z13tz_ipm_kernelILi1ELi1ELi4EEv9IpmParams:/{f=1} f{print} f&&/^\.Lfunc_end/{exit}' /tmp/k_small.s > /tmp/di.s
echo "lines $(wc -l < /tmp/di.s) scratch_load $(grep -c scratch_load /tmp/di.s) scratch_store $(grep -c scratch_store /tmp/di.s) writelane $(grep -c v_writelane /tmp/di.s) readlane $(grep -c v_readlane /tmp/di.s) barriers $(grep -c s_barrier /tmp/di.s)"
awk '/^_Z13tz_ipm_kernelILi1ELi1ELi4EEv9IpmParams:/{f=1} f&&/; (NumSgprs|NumVgprs|ScratchSize|Occupancy|codeLenInByte)/{print}' /tmp/k_small.s | head -5
