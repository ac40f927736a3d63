#!/bin/bash
# A/B of library builds on the 2M-triangle mesh workload (64 spp): M samples/s and per-kernel microseconds.
# usage: tools/ab_mesh.sh [bench args, e.g. --bvh device] -- variant...   (variants live in path-tracing_amd/lib/variants)
args=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args="$args $1"; shift; done
shift
for v in default "$@"; do
  if [ "$v" = default ]; then unset PT_AMD_LIB; else export PT_AMD_LIB=$GRAFT_REPO_ROOT/path-tracing_amd/lib/variants/libptamd_$v.so; fi
  r=$(timeout -k 10 300 python bench.py --workload mesh1000 --steps 1 --warmup 1 --sqrt-spp 8 --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f M samples/s  %.2f ms/step' % (d['value']/1e6, d['ms_per_step']), {k[:8]: (round(x['avg_us'],1), int(x['launches_per_step'])) for k,x in d['kernels'].items()})")
  echo "$v: $r"
done
