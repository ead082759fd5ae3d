#!/bin/bash
# usage: scripts/sweep_tune.sh VAR "v1 v2 ..."   -- C2 bench (no closest-hit part) once per value of the environment variable VAR
mkdir -p gpurun_out
for v in $2; do
  export $1=$v
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-closest --no-other-configs > /tmp/sw.log 2>/dev/null || { echo "FAILED $v"; exit 1; }
  echo "$1=$v $(python scripts/show_bench.py /tmp/sw.log | cut -c1-130)"
done
