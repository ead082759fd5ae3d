#!/bin/bash
# usage: scripts/ab_variants.sh <name>...  -- C2 bench + closest-hit micro-benchmark with variants/libmipt_<name>.so ("base" = the shipped library)
mkdir -p gpurun_out
for n in "$@"; do
  if [ "$n" = base ]; then unset MIPT_LIB; else export MIPT_LIB=$PWD/variants/libmipt_$n.so; fi
  echo "== $n"
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/ab_$n.log 2> gpurun_out/ab_$n.err || { echo "FAILED $n"; tail -5 gpurun_out/ab_$n.err; exit 1; }
  python scripts/show_bench.py gpurun_out/ab_$n.log | cut -c1-140
done
