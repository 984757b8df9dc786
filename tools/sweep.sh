#!/bin/bash
# tuning sweep on the GPU box (experiments only): bench goes_hrit under several env settings
run() { echo "== $*"; env "$@" python bench.py --steps 3 --warmup 1 --cpu-samples 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('Msps',d['value'],'ms',d['ms_per_step'],'check',d['check']['cadus_matching_transmitted'],'stats',d['demod_stats'])
print('  '+' '.join(f\"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}/{v['launches_per_step']:.0f}\" for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:8]))
"; }
for s in "$@"; do run $s; done
