#!/bin/bash
# tools/spp_sweep.sh for several library builds: tools/spp_sweep_ab.sh WORKLOAD "sqrt_spp..." variant...
wl=$1; spps=$2; shift 2
for v in default "$@"; do
  if [ "$v" = default ]; then unset PT_AMD_LIB; else export PT_AMD_LIB=$GRAFT_REPO_ROOT/path-tracing_amd/lib/variants/libptamd_$v.so; fi
  echo "== $v"
  bash tools/spp_sweep.sh $wl $spps
done
