#!/bin/bash
# sweep of the scheduling knobs used for the bounce-ray launches (MIPT_TUNE) and for the coherent ones (MIPT_TUNE_COHERENT)
for t in ${TUNES:-"32,28,4"}; do
  for c in ${CTUNES:-"32,28,4"}; do
    export MIPT_TUNE=$t MIPT_TUNE_COHERENT=$c
    timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-closest --no-cpu-baseline | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bounce=$t coherent=$c', d['value']/1e6, d['roofline']['Mrays_per_s_in_kernel'])"
  done
done
