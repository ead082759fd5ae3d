#!/bin/bash
# A/B of library builds on the bidirectional config: scripts/ab_c3.sh variants/libmipt_*.so
cp cuda_path_tracer_amd/libmipt.so /tmp/keep.so
for f in "$@" /tmp/keep.so; do
  cp "$f" cuda_path_tracer_amd/libmipt.so
  echo "== $f"; timeout -k 10 120 python scripts/sweep_c3.py 16 2>&1 | grep samples_per
done
cp /tmp/keep.so cuda_path_tracer_amd/libmipt.so
