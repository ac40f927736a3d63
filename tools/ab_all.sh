#!/bin/bash
# The four workloads, the library as built against variants: tools/ab_all.sh variant...   (on the GPU box)
for wl in cornell frames textured mesh1000; do
  bash tools/ab_bench.sh $wl -- "$@" || exit 1
done
