#!/bin/bash
# Re-measure what profiles/ holds for the current build (run on the GPU box; results land in gpurun_out/refresh,
# copy what is to be kept into profiles/).  Every step has its own time limit and a failure ends the script.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/default" -o default -- \
    python3 bench.py > "$OUT/r01_bench_default.json" 2> "$OUT/default.err"
echo default done
timeout -k 10 300 python3 bench.py --workload textured --steps 1 --warmup 1 > "$OUT/r01_bench_textured.json" 2> "$OUT/textured.err"
echo textured done
timeout -k 10 300 python3 bench.py --workload frames --steps 1 --warmup 1 > "$OUT/r01_bench_frames.json" 2> "$OUT/frames.err"
echo frames done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mesh" -o mesh -- \
    python3 bench.py --workload mesh1000 --steps 1 --warmup 1 > "$OUT/r01_bench_mesh1000.json" 2> "$OUT/mesh.err"
echo mesh done
timeout -k 10 300 python3 bench.py --workload mesh1000 --bvh device --steps 1 --warmup 1 > "$OUT/r01_bench_mesh1000_devicebvh.json" 2> "$OUT/meshdev.err"
echo mesh device done
bash tools/collect_traffic.sh cornell > "$OUT/traffic_cornell.out" 2>&1
echo traffic cornell done
bash tools/collect_traffic.sh mesh1000 > "$OUT/traffic_mesh.out" 2>&1
echo traffic mesh done
