#!/bin/bash
# Copy the results of tools/refresh_profiles.sh (merged back into gpurun_out/) into profiles/ under the round's prefix.
set -e
P=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$ROOT/gpurun_out/refresh
cp "$R"/${P}_bench_*.json "$ROOT/profiles/"
cp "$R"/default/default_kernel_stats.csv "$ROOT/profiles/${P}_bench_default_kernel_stats.csv"
cp "$R"/mesh/mesh_kernel_stats.csv "$ROOT/profiles/${P}_bench_mesh1000_kernel_stats.csv"
cp "$R"/${P}_pmc_*.txt "$R"/${P}_bvh_build_times.txt "$ROOT/profiles/" 2>/dev/null || true
for WL in cornell mesh1000 mesh2000 textured frames; do
    [ -s "$ROOT/gpurun_out/traffic_$WL.json" ] && cp "$ROOT/gpurun_out/traffic_$WL.json" "$ROOT/profiles/traffic_$WL.json"
done
ls -la "$ROOT/profiles" | tail -20
