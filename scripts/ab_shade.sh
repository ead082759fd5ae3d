#!/bin/bash
# usage: scripts/ab_shade.sh <name>...  -- rocprofv3 kernel trace of two C2 steps per library variant; prints k_shade per vertex
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = base ]; then unset MIPT_LIB; else export MIPT_LIB=$GRAFT_REPO_ROOT/variants/libmipt_$n.so; fi
  rm -rf /tmp/ks_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-pmc --no-closest > /tmp/ks_$n.log 2>&1 || { echo "FAILED $n"; tail -5 /tmp/ks_$n.log; exit 1; }
  python3 - $n <<'PY'
import csv,glob,sys
n=sys.argv[1]
f=glob.glob(f"/tmp/ks_{n}/**/*kernel_trace.csv",recursive=True)[0]
sh=[];tot={}
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    key="k_shade" if "k_shade" in k else "k_extend<0>" if "k_extendILi0" in k or "k_extend<0>" in k else "k_resolve" if "k_resolve" in k else "sort" if "rocprim" in k else "other"
    tot[key]=tot.get(key,0)+d
    if key=="k_shade": sh.append(d)
print("%-12s k_shade per vertex (us):"%n," ".join("%.0f"%x for x in sh[:6]),"| totals ms:"," ".join("%s %.1f"%(k,v/1e3) for k,v in sorted(tot.items())))
PY
done
