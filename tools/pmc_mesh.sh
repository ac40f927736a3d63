#!/bin/bash
# SQ counters of the mesh workload (64 spp) for library variants: tools/pmc_mesh.sh [bench args] -- variant...
# (one rocprofv3 --pmc run per variant; summaries in gpurun_out/pmc_mesh_<variant>.txt)
args=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args="$args $1"; shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = default ]; then unset PT_AMD_LIB; else export PT_AMD_LIB=$GRAFT_REPO_ROOT/path-tracing_amd/lib/variants/libptamd_$v.so; fi
  rm -rf gpurun_out/pmc_$v
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU \
      --output-format csv -d gpurun_out/pmc_$v -o p -- python3 bench.py --workload mesh1000 --steps 1 --warmup 0 --sqrt-spp 8 --no-cpu-baseline $args > /dev/null 2>&1 || exit 1
  python3 tools/pmc_summary.py gpurun_out/pmc_$v/p_counter_collection.csv > gpurun_out/pmc_mesh_$v.txt
  echo "== $v"; head -4 gpurun_out/pmc_mesh_$v.txt
  rm -rf gpurun_out/pmc_$v
done
