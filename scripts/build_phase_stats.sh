#!/bin/bash
# Developer build of libmipt with the per-phase lane-use table of the extend kernel (variants/libmipt_stats.so).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
cd "$HERE/../cuda_path_tracer_amd/csrc"
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-bitwise-instead-of-logical -pthread -DMIPT_PHASE_STATS"
mkdir -p /tmp/ps_build ../../variants
hipcc --offload-arch=gfx950 $F -c api.hip -o /tmp/ps_build/api.o
hipcc --offload-arch=gfx950 $F -c wavefront.hip -o /tmp/ps_build/wavefront.o
hipcc --offload-arch=gfx950 -shared -fPIC -pthread /tmp/ps_build/api.o /tmp/ps_build/wavefront.o scene.o assets.o kd_build.o kd_build_gpu.o pack_gpu.o -o ../../variants/libmipt_stats.so
# ... and the build that clocks the sections of the persistent loop instead (variants/libmipt_clocks.so)
bash "$HERE/build_variant.sh" clocks -DMIPT_PHASE_CLOCKS
