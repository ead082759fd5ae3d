#!/bin/bash
# usage: scripts/try_variants.sh variants/libmipt_*.so  -- runs the C2 bench with each library build in turn
set -e
mkdir -p gpurun_out
cp cuda_path_tracer_amd/libmipt.so /tmp/libmipt_keep.so
for f in "$@"; do
  cp "$f" cuda_path_tracer_amd/libmipt.so
  echo "== $f" | tee -a gpurun_out/variants.log
  timeout -k 10 200 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-closest > /tmp/v.log 2>&1
  python scripts/show_bench.py /tmp/v.log | tee -a gpurun_out/variants.log
done
cp /tmp/libmipt_keep.so cuda_path_tracer_amd/libmipt.so
