#!/bin/bash
# one-dimensional sweeps of the persistent extend kernel's scheduling knobs (MIPT_TUNE=refill,batch,cap)
for t in ${TUNES:-"12,24,4" "12,24,2" "12,24,3" "12,24,6" "12,24,8" "8,24,4" "20,24,4" "12,16,4" "12,32,4"}; do
  export MIPT_TUNE=$t
  a=$(timeout -k 10 120 python scripts/prof_closest.py incoherent 23 3 | tail -1 | awk '{print $NF}')
  b=$(timeout -k 10 120 python scripts/prof_closest.py coherent 23 3 | tail -1 | awk '{print $NF}')
  c=$(timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-closest --no-cpu-baseline | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['roofline']['Mrays_per_s_in_kernel'])")
  echo "TUNE=$t incoherent=$a coherent=$b C2=$c"
done
