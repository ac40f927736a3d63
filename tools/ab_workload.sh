#!/bin/bash
# A/B of library builds on one bench workload: tools/ab_workload.sh WORKLOAD "BENCH ARGS" variant...
wl=$1; shift; extra=$1; shift
for v in default "$@"; do
  if [ "$v" = default ]; then unset PT_AMD_LIB; else export PT_AMD_LIB=$GRAFT_REPO_ROOT/path-tracing_amd/lib/variants/libptamd_$v.so; fi
  r=$(timeout -k 10 300 python bench.py --workload $wl --steps 1 --warmup 1 $extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f M samples/s  %.2f ms/step' % (d['value']/1e6, d['ms_per_step']), {k[:5]: (round(x['avg_us'],1), int(x['launches_per_step'])) for k,x in d['kernels'].items()})")
  echo "$v: $r"
done
