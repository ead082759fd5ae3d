#!/bin/bash
# usage: scripts/prof_config.sh C3|C4|C5|C1  -- rocprofv3 kernel stats of one of the other BASELINE configurations
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc_$1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$1 -- python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py $1 > /tmp/pc_$1.log 2>&1 || { tail -5 /tmp/pc_$1.log; exit 1; }
grep '^{' /tmp/pc_$1.log | cut -c1-200
python3 $GRAFT_REPO_ROOT/scripts/kernel_stats_summary.py /tmp/pc_$1
