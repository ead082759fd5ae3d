#!/bin/bash
# usage: scripts/prof_bench.sh <tag> [bench args]  -- rocprofv3 kernel stats of a short C2 bench run; copies the csv to gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pbn_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbn_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-closest "$@" > /tmp/pbn_$tag.log 2>&1 || { tail -5 /tmp/pbn_$tag.log; exit 1; }
f=$(find /tmp/pbn_$tag -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,re,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    n=r["Name"]; m=re.search(r'(k_\w+(<\(?\w*\)?\d>)?)',n); print("  %-24s calls %4s avg %9.1f us total %9.2f ms %5s%%"%(m.group(1) if m else n[:24],r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6,r["Percentage"][:5]))
PY
