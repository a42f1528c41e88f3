#!/bin/bash
# Production-schedule cost of the phases of the fused step: builds of the bench-problem kernel that execute ONE phase TZ_DUP_N = 2 more
# times (same results; tools/devlib.sh dupK -DTZ_DUP=K -DTZ_DUP_N=2) against the base build, three bench runs each (tools/ab.sh).
# On the GPU box: tools/dup_phases.sh > gpurun_out/dup.txt
cd "$(dirname "$0")/.."
libs=(tzddpc_amd/lib/ab/base.so)
for k in 1 2 3 4 5 6 7 9 10; do libs+=(tzddpc_amd/lib/ab/dup$k.so); done
tools/ab.sh "${libs[@]}"
