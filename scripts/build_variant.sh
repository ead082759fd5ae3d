#!/bin/bash
# usage: scripts/build_variant.sh <name> [extra hipcc flags, e.g. -DFOO=1]  -> variants/libmipt_<name>.so (A/B timing with MIPT_LIB / scripts/ab_variants.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/../cuda_path_tracer_amd/csrc"
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-bitwise-instead-of-logical -pthread $*"
B=/tmp/variant_$name
mkdir -p $B ../../variants
hipcc --offload-arch=gfx950 $F -c api.hip -o $B/api.o &
hipcc --offload-arch=gfx950 $F -c wavefront.hip -o $B/wavefront.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -pthread $B/api.o $B/wavefront.o scene.o assets.o kd_build.o kd_build_gpu.o pack_gpu.o -o ../../variants/libmipt_$name.so
