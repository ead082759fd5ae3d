#!/bin/bash
# usage: scripts/prof_binary.sh <tag> <binary> [args...]  -- rocprofv3 kernel stats of a standalone micro-benchmark
tag=$1; shift
bin=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$tag -- $bin "$@" > /tmp/pb_$tag.log 2>&1 || { tail -5 /tmp/pb_$tag.log; exit 1; }
python3 $GRAFT_REPO_ROOT/scripts/kernel_stats_summary.py /tmp/pb_$tag
