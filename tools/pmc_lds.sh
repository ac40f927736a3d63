#!/bin/bash
# LDS counters of a workload's kernels (one rocprofv3 --pmc run): how busy the LDS is and how many of its cycles
# are bank conflicts, beside the vector instruction counters: tools/pmc_lds.sh WORKLOAD -> gpurun_out/pmc_lds_<workload>.txt
WL=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmclds_$WL
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --output-format csv -d gpurun_out/pmclds_$WL -o p -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/pmc_lds_$WL.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmclds_$WL/p_counter_collection.csv > gpurun_out/pmc_lds_$WL.txt
rm -rf gpurun_out/pmclds_$WL
head -6 gpurun_out/pmc_lds_$WL.txt
