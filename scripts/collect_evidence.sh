#!/bin/bash
# One GPU call that refreshes everything profiles/ cites: default bench, rocprofv3 kernel stats of the bench command,
# PMC passes (bench workload + closest-hit batches), the other BASELINE configurations.  Outputs under gpurun_out/ev/.
set -e
O=gpurun_out/ev; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.log 2>&1
grep '^{' $O/bench_default.log | tail -1 > $O/bench_default.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-closest > $O/bench_under_rocprof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
grep '^{' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
bash scripts/pmc_bench.sh > $O/pmc_bench.log 2>&1 && cp gpurun_out/pmc_bench/summary.json $O/pmc_bench_c2.json
bash scripts/pmc.sh ev_incoh incoherent > $O/pmc_incoherent_4M.txt 2>&1
bash scripts/pmc.sh ev_coh coherent > $O/pmc_coherent_4M.txt 2>&1
timeout -k 10 900 python scripts/bench_configs.py C1 C3 C4 C5 > $O/other_configs.log 2>&1
grep '^{' $O/other_configs.log > $O/other_configs.jsonl
echo done
