#!/bin/bash
# A/B of library builds on one workload: M samples/s and per-kernel microseconds.
# usage: tools/ab_bench.sh WORKLOAD [bench args] -- variant...   (variants live in path-tracing_amd/lib/variants)
wl=$1; shift
args=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args="$args $1"; shift; done
shift
for v in default "$@"; do
  if [ "$v" = default ]; then unset PT_AMD_LIB; else export PT_AMD_LIB=$GRAFT_REPO_ROOT/path-tracing_amd/lib/variants/libptamd_$v.so; fi
  # (a build that fails ends the comparison: nothing more is started on the GPU after a failure)
  r=$(set -o pipefail; timeout -k 10 300 python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f M samples/s  %.2f ms/step' % (d['value']/1e6, d['ms_per_step']), {k[:8]: (round(x['avg_us'],1), int(x['launches_per_step'])) for k,x in d['kernels'].items()}, 'rmse ok', (d.get('rmse') or {}).get('within_bound'))")
  rc=$?
  echo "$wl $v: $r"
  if [ $rc -ne 0 ] || [ -z "$r" ]; then echo "$wl $v failed: stopping"; exit 1; fi
done
