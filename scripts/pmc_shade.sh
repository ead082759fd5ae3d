#!/bin/bash
# PMC passes for the shading kernels of the bench workload (k_shade / k_resolve); summary on stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_shade
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_shade/p$i -- python bench.py --steps 1 --warmup 0 --no-closest --no-cpu-baseline --no-pmc > gpurun_out/pmc_shade/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmc_shade/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for k in ("k_shade", "k_resolve", "k_raygen"):
            if k in row["Kernel_Name"]:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-26s n=%d first=%.4g mean=%.4g" % (c, len(v), v[0], sum(v) / len(v)))
PY
