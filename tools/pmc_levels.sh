#!/bin/bash
# usage (on the GPU box): tools/pmc_levels.sh <tag>   -- average latencies of the timed launch from the SQ level counters, one
# rocprofv3 --pmc pass per group (TZ_LIB selects the library): vector memory, LDS, scalar memory, instruction fetch; busy cycles by unit
tag=$1
tools/pmc_pass.sh ${tag}_vmem SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
tools/pmc_pass.sh ${tag}_lds SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
tools/pmc_pass.sh ${tag}_smem SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA
tools/pmc_pass.sh ${tag}_ifetch SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY
tools/pmc_pass.sh ${tag}_act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY
