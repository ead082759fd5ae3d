#!/bin/bash
# ONE GPU call that regenerates everything profiles/r3_* cites (run at the final commit of a round):
#   bash scripts/collect_profiles.sh      (on the GPU box, from the repository root; ~3 minutes)
# -> gpurun_out/r3/: bench_default.json (the driver's command, PMC passes live), pmc_bench.json (the counters behind its
#    roofline objects), bench_under_rocprof.json + kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same workload:
#    k_extend's average duration must agree with roofline.avg_launch_ms), other_configs.jsonl (C1, C3, C4, C5),
#    phase_lane_use.txt (developer build: lane use per phase of the persistent kernel, refills per cursor word),
#    phase_clocks.txt (developer build: shader clocks per section of the persistent loop), check_unfused.txt (the as-written
#    ray_at on both sides), kd_sort_check.txt (kd_sort.h against std::sort and rocPRIM),
#    bin_sort_bench.txt (queue binning of queue_sort.h against rocPRIM's radix sort, 45 M pairs, three key distributions).
# Copy what is to be judged into profiles/ with the r3_ prefix.
set -e
O=gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --save-pmc $O/pmc_bench.json > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-pmc --no-cpu-baseline --no-closest --no-other-configs > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof
timeout -k 10 900 python scripts/bench_configs.py C1 C3 C4 C4L C5 > $O/other_configs.log 2>&1
grep '^{' $O/other_configs.log > $O/other_configs.jsonl
if [ -f variants/libmipt_stats.so ]; then
  MIPT_LIB=variants/libmipt_stats.so timeout -k 10 300 python scripts/phase_lane_use.py > $O/phase_lane_use.txt 2>&1
fi
if [ -f variants/libmipt_clocks.so ]; then
  MIPT_LIB=$PWD/variants/libmipt_clocks.so timeout -k 10 300 python scripts/phase_lane_use.py > $O/phase_clocks.txt 2>&1
fi
if [ -f variants/libmipt_unfused.so ]; then
  MIPT_LIB=$PWD/variants/libmipt_unfused.so timeout -k 10 300 python scripts/check_unfused.py > $O/check_unfused.txt 2>&1
fi
if [ -x scripts/micro/kd_sort_check ]; then
  timeout -k 10 120 scripts/micro/kd_sort_check > $O/kd_sort_check.txt 2>&1
fi
if [ -x scripts/micro/bin_sort_bench ]; then
  timeout -k 10 120 scripts/micro/bin_sort_bench > $O/bin_sort_bench.txt 2>&1
fi
echo done
