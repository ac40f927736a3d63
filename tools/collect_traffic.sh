#!/bin/bash
# HBM traffic per kernel launch of bench.py's workload: two rocprofv3 --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), each a run of its own over exactly the launches of one timed step
# (bench.py --pmc-pass: no instrumented renders, no warm-up), then tools/traffic_json.py merges them into
# gpurun_out/traffic_<workload>.json (copy it to profiles/ to have bench.py report it).
#   tools/collect_traffic.sh <workload> [bench args...]
set -e
WL=${1:-cornell}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/traffic_$WL
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- \
    python3 bench.py --workload "$WL" --steps 1 --warmup 0 --pmc-pass "$@" > "$OUT/fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- \
    python3 bench.py --workload "$WL" --steps 1 --warmup 0 --pmc-pass "$@" > "$OUT/write.log" 2>&1
python3 tools/traffic_json.py "$WL" "$OUT/fetch/fetch_counter_collection.csv" "$OUT/write/write_counter_collection.csv" \
    > "$ROOT/gpurun_out/traffic_$WL.json"
cat "$ROOT/gpurun_out/traffic_$WL.json"
