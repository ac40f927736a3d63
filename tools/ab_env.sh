#!/bin/bash
# A/B of environment switches on one workload with the library as built: M samples/s and per-kernel microseconds.
# usage: tools/ab_env.sh WORKLOAD [bench args] -- "" "PT_X=1" "PT_Y=2 PT_Z=3" ...   (each argument = one environment)
wl=$1; shift
args=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args="$args $1"; shift; done
shift
for e in "$@"; do
  # (a run that fails ends the comparison: nothing more is started on the GPU after a failure)
  r=$(set -o pipefail; env $e timeout -k 10 300 python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f M samples/s  %.2f ms/step' % (d['value']/1e6, d['ms_per_step']), {k[:8]: (round(x['avg_us'],1), int(x['launches_per_step'])) for k,x in d['kernels'].items()}, 'rows/sample %.1f' % d['executed_walk_per_sample']['box_rows'], 'rmse ok', (d.get('rmse') or {}).get('within_bound'))")
  rc=$?
  echo "$wl [$e]: $r"
  if [ $rc -ne 0 ] || [ -z "$r" ]; then echo "$wl [$e] failed: stopping"; exit 1; fi
done
