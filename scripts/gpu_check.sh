#!/bin/bash
# GPU tests + C2 bench + closest-hit microbench [+ phase table with the developer build]; log name = $1
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/$1.log 2>&1
python scripts/show_bench.py gpurun_out/$1.log
if [ -n "$2" ]; then
  cp variants/libmipt_stats.so cuda_path_tracer_amd/libmipt.so
  timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/$1_phase.log 2>&1 || true
  grep -A11 "phase stats" gpurun_out/$1_phase.log | sed -n 1,24p
fi
