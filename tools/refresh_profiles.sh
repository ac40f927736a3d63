#!/bin/bash
# Re-measure what profiles/ holds for the current build (run on the GPU box; results land in gpurun_out/refresh,
# tools/keep_profiles.sh copies what is to be kept into profiles/ under the round's prefix).  Every step has its
# own time limit and a failure ends the script.
#   tools/refresh_profiles.sh [prefix, default r02]
set -e
P=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
# the counter passes first: the benchmark lines below quote them from profiles/
for WL in cornell mesh1000 textured frames; do
    bash tools/collect_traffic.sh $WL > "$OUT/traffic_$WL.out" 2>&1
    cp "$ROOT/gpurun_out/traffic_$WL.json" "$ROOT/profiles/traffic_$WL.json"
    echo traffic $WL done
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/default" -o default -- \
    python3 bench.py > "$OUT/${P}_bench_default.json" 2> "$OUT/default.err"
echo default done
timeout -k 10 300 python3 bench.py --workload textured --steps 1 --warmup 1 > "$OUT/${P}_bench_textured.json" 2> "$OUT/textured.err"
echo textured done
timeout -k 10 300 python3 bench.py --workload frames --steps 1 --warmup 1 > "$OUT/${P}_bench_frames.json" 2> "$OUT/frames.err"
echo frames done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mesh" -o mesh -- \
    python3 bench.py --workload mesh1000 --steps 1 --warmup 1 > "$OUT/${P}_bench_mesh1000.json" 2> "$OUT/mesh.err"
echo mesh done
timeout -k 10 300 python3 bench.py --workload cornell --bvh file --steps 1 --warmup 1 > "$OUT/${P}_bench_cornell_filetree.json" 2> "$OUT/cornellfile.err"
echo cornell with the tree of the scene file done
