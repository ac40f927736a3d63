#!/bin/bash
# The N-rank benchmark rehearsed on a one-GPU machine (every rank on GPU 0, exchange over gloo), all workloads.
#   tools/rehearse_ranks.sh [ranks, default 2]
set -e
N=${1:-2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/rehearse
mkdir -p "$OUT"
cd "$ROOT"
for w in cornell textured frames3 mesh1000; do
    PT_BENCH_REHEARSAL=1 timeout -k 10 280 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
        --master-port 29511 bench.py --gpus $N --steps 1 --warmup 1 --workload $w --no-cpu-baseline > "$OUT/$w.json" 2> "$OUT/$w.err"
    python3 - "$OUT/$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], d["n_gpus"], "ranks", round(d["value"] / 1e6, 1), "M samples/s", d["config"].get("sharding"),
      "rmse ok:", (d.get("rmse") or {}).get("within_bound"), flush=True)
PY
done
