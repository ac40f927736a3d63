#!/bin/bash
# Quick variant of the library that differs only in pt_wavefront.hip's compile flags (the other objects are taken from
# the last `make` in path-tracing_amd/csrc): tools/build_wavefront_variant.sh NAME -DPT_SOMETHING=1 ...
set -e
name=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/path-tracing_amd/lib/variants
mkdir -p "$OUT" /tmp/ptw_$name
cd "$ROOT/path-tracing_amd/csrc"
/opt/rocm/bin/hipcc -std=c++17 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -I../../include -I. "$@" --offload-arch=gfx950 -c pt_wavefront.hip -o /tmp/ptw_$name/pt_wavefront.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 pt_host.o pt_kernels.o /tmp/ptw_$name/pt_wavefront.o pt_bvh_device.o pt_ingest.o pt_multi.o pt_reindex.o -o "$OUT/libptamd_$name.so"
echo built $name
