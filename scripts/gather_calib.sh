#!/bin/bash
# Calibrates FETCH_SIZE / TCC_MISS for gather-shaped reads (scripts/micro/gather_calib.hip).  Run on the GPU box:
#   bash scripts/gather_calib.sh      -> gpurun_out/gather_calib/summary.txt  (copied to profiles/r2_gather_calibration.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/gather_calib; mkdir -p $O
BIN=scripts/micro/gather_calib
[ -x $BIN ] || { echo "build $BIN first (hipcc --offload-arch=gfx950 -O3 $BIN.hip -o $BIN)"; exit 1; }
: > $O/summary.txt
for mode in 0 1 2 3 4; do
  timeout -k 5 60 $BIN $mode >> $O/summary.txt || exit 1
  p=0
  for ctrs in "FETCH_SIZE TCC_MISS_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    p=$((p+1))
    timeout -s KILL 90 rocprofv3 --pmc $ctrs --output-format csv -d $O/m${mode}p$p -- $BIN $mode > $O/m${mode}p$p.log 2>&1 || { echo "mode $mode pass $p failed" >> $O/summary.txt; tail -3 $O/m${mode}p$p.log >> $O/summary.txt; }
  done
  python3 - $O $mode >> $O/summary.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
root, mode = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob("%s/m%sp*/**/*counter_collection.csv" % (root, mode), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_calib" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
n = 1 << 24
lines = n / 8 if mode == "0" else n
for k in sorted(acc):
    v = sum(acc[k]) / len(acc[k])
    extra = "  = %.1f B per distinct line" % (v * 1024 / lines) if k == "FETCH_SIZE" else "  = %.3f per distinct line" % (v / lines)
    print("    %-24s mean %.6g over %d launches%s" % (k, v, len(acc[k]), extra))
PY
done
cat $O/summary.txt
