#!/bin/bash
# like sweep.sh for another workload: tools/sweep_wl.sh <workload> "ENV=.. ENV=.." ...
WL=$1; shift
for s in "$@"; do echo "== $WL $s"; env $s python bench.py --workload $WL --steps 2 --warmup 1 --cpu-samples 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('Msps',d['value'],'ms',d['ms_per_step'],'check',d['check']['cadus_matching_transmitted'],'/',d['check']['transmitted'],'stats',d['demod_stats'])
print('  '+' '.join(f\"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}/{v['launches_per_step']:.0f}\" for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:8]))
"; done
