#!/bin/bash
OUT=gpurun_out/dbg2; mkdir -p $OUT
for v in "A=1" "SDHIP_WINDOW_GATHER=0" "SDHIP_FEC_BATCH=20000"; do
  for cs in 0 4000000; do
    env $v python bench.py --frames 42000 --cpu-samples $cs --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v cpu=$cs', d['check'], d['cadu_parity'])"
  done
done
