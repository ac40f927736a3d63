#!/bin/bash
# Build a variant of the library with extra compiler flags into path-tracing_amd/lib/variants/libptamd_NAME.so
#   tools/build_variant.sh NAME -DPT_SOMETHING=1 ...
#   SRC=/path/to/other/checkout tools/build_variant.sh NAME     (sources from another tree, e.g. a git worktree)
set -e
name=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC:-$ROOT}
OUT=$ROOT/path-tracing_amd/lib/variants
mkdir -p "$OUT" /tmp/ptv_$name
cd "$SRC/path-tracing_amd/csrc"
FLAGS="-std=c++17 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-result -I../../include -I. $*"
objs=""
for f in *.hip; do
  /opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 -c $f -o /tmp/ptv_$name/${f%.hip}.o &
  objs="$objs /tmp/ptv_$name/${f%.hip}.o"
done
/opt/rocm/bin/hipcc $FLAGS -x c++ -c pt_host.cpp -o /tmp/ptv_$name/pt_host.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/ptv_$name/pt_host.o $objs -o "$OUT/libptamd_$name.so"
echo built $name
