#!/bin/bash
# Build a variant of the library with extra compiler flags into path-tracing_amd/lib/variants/libptamd_NAME.so
#   tools/build_variant.sh NAME -DPT_SOMETHING=1 ...
set -e
name=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/path-tracing_amd/csrc"
mkdir -p ../lib/variants /tmp/ptv_$name
FLAGS="-std=c++17 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -I../../include -I. $*"
/opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 -c pt_wavefront.hip -o /tmp/ptv_$name/pt_wavefront.o &
/opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 -c pt_kernels.hip -o /tmp/ptv_$name/pt_kernels.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 pt_host.o pt_bvh_device.o pt_ingest.o /tmp/ptv_$name/pt_kernels.o /tmp/ptv_$name/pt_wavefront.o -o ../lib/variants/libptamd_$name.so
echo built $name
