#!/bin/bash
# usage: scripts/pmc.sh <tag> <kind> -- separate rocprofv3 --pmc passes (counter limits per block respected)
set -e
tag=$1; kind=${2:-incoherent}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
            "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag/p$i -- python scripts/prof_closest.py $kind 22 2 > gpurun_out/pmc_$tag/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_$tag/p$i.log; }
done
python scripts/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/pmc_$tag/summary.txt
cat gpurun_out/pmc_$tag/summary.txt
