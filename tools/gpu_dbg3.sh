#!/bin/bash
for v in "SDHIP_VIT2=0"; do
  env $v python bench.py --cpu-samples 0 --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['check'])"
done
