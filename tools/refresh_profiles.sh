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
for WL in cornell mesh1000 mesh2000 textured frames; do
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
timeout -k 10 400 python3 bench.py --workload mesh2000 --steps 1 --warmup 1 > "$OUT/${P}_bench_mesh2000.json" 2> "$OUT/mesh2000.err"
echo mesh2000 done
timeout -k 10 300 python3 bench.py --workload cornell --bvh file --steps 1 --warmup 1 > "$OUT/${P}_bench_cornell_filetree.json" 2> "$OUT/cornellfile.err"
echo cornell with the tree of the scene file done

timeout -k 10 300 python3 bench.py --workload textured --bvh file --steps 1 --warmup 1 > "$OUT/${P}_bench_textured_filetree.json" 2> "$OUT/texturedfile.err"
echo textured with the tree of the scene file done
# instruction-issue counters of every workload, cache counters of the two meshes (each its own rocprofv3 --pmc run)
for WL in cornell frames textured mesh1000 mesh2000; do
    bash tools/pmc_sq.sh $WL > /dev/null 2>&1 && cp "$ROOT/gpurun_out/pmc_sq_$WL.txt" "$OUT/${P}_pmc_sq_$WL.txt"
    echo sq counters $WL done
done
for WL in mesh1000 mesh2000; do
    bash tools/pmc_cache.sh $WL > /dev/null 2>&1 && cp "$ROOT/gpurun_out/pmc_cache_${WL}_tcp.txt" "$OUT/${P}_pmc_cache_${WL}_tcp.txt" \
        && cp "$ROOT/gpurun_out/pmc_cache_${WL}_tcc.txt" "$OUT/${P}_pmc_cache_${WL}_tcc.txt"
    echo cache counters $WL done
done
timeout -k 10 200 python3 tools/bench_bvh_build.py 150 500 1000 2000 > "$OUT/${P}_bvh_build_times.txt" 2>&1
echo tree build times done
