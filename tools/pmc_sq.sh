#!/bin/bash
# Instruction-issue counters of a workload's kernels (one rocprofv3 --pmc run: 8 SQ counters + GRBM_GUI_ACTIVE for the
# clock): tools/pmc_sq.sh WORKLOAD [bench args...] -> gpurun_out/pmc_sq_<workload>.txt
WL=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcsq_$WL
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE \
    --output-format csv -d gpurun_out/pmcsq_$WL -o p -- python3 bench.py --workload $WL --steps 1 --warmup 0 --pmc-pass "$@" > /dev/null 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmcsq_$WL/p_counter_collection.csv > gpurun_out/pmc_sq_$WL.txt
rm -rf gpurun_out/pmcsq_$WL
head -12 gpurun_out/pmc_sq_$WL.txt
