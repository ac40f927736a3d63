#!/bin/bash
# All four workloads of the benchmark with the given extra arguments, one summary line each.
#   tools/bench_all.sh <tag> [bench.py arguments...]      results in gpurun_out/bench_all/<tag>_<workload>.json
set -e
TAG=$1
shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/bench_all
mkdir -p "$OUT"
cd "$ROOT"
for w in cornell frames textured mesh1000; do
    timeout -k 10 300 python3 bench.py --workload $w --steps 1 --warmup 1 "$@" > "$OUT/${TAG}_$w.json" 2> "$OUT/${TAG}_$w.err"
    python3 - "$OUT/${TAG}_$w.json" $w <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.1f M samples/s" % (d["value"] / 1e6), "build ms", d["config"].get("bvh_build_ms"),
      "rmse ok", (d.get("rmse") or {}).get("within_bound"), {k: round(v["avg_us"]) for k, v in d["kernels"].items()}, flush=True)
PY
done
