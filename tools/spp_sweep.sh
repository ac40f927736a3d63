#!/bin/bash
# Rate against samples per pixel (what one of N GPUs renders under strong scaling): tools/spp_sweep.sh WORKLOAD sqrt_spp...
wl=$1; shift
for q in "$@"; do
  r=$(timeout -k 10 300 python bench.py --workload $wl --steps 2 --warmup 1 --sqrt-spp $q --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; t=[v for n,v in k.items() if n.startswith('trace') and 'Retry' not in n][0]; print('%.1f M samples/s  %.2f ms/step' % (d['value']/1e6, d['ms_per_step']), 'iterations', int(t['launches_per_step']), 'trace avg us', round(t['avg_us'],1))")
  echo "$wl sqrt_spp $q: $r"
done
