#!/bin/bash
# Pool size sweep on the default workload: tools/sweep_slots.sh SLOTS...
for s in "$@"; do
  r=$(PT_WAVEFRONT_SLOTS=$s timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step'],1), {k[:5]:(round(v['avg_us'],1), int(v['launches_per_step'])) for k,v in d['kernels'].items()})")
  echo "$s: $r"
done
