#!/bin/bash
# Cache-level counters of a workload's kernels: one rocprofv3 --pmc run for the vector L1 (TCP), one for the L2 (TCC).
#   tools/pmc_cache.sh WORKLOAD [bench args...]    -> gpurun_out/pmc_cache_<workload>_{tcp,tcc}.txt
WL=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { # name counters...
  local name=$1; shift
  rm -rf gpurun_out/pmcc_$name
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcc_$name -o p -- \
      python3 bench.py --workload $WL --steps 1 --warmup 0 --sqrt-spp 8 --no-cpu-baseline $EXTRA > /dev/null 2>&1 || return 1
  python3 tools/pmc_summary.py gpurun_out/pmcc_$name/p_counter_collection.csv > gpurun_out/pmc_cache_${WL}_$name.txt
  rm -rf gpurun_out/pmcc_$name
  head -6 gpurun_out/pmc_cache_${WL}_$name.txt
}
EXTRA="$*"
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum && \
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
