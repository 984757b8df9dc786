#!/bin/bash
# quick A/B: GPU tests (optionally a subset via $TESTS) + short bench lines for the workloads given (kernel table, top entries)
python -m pytest ${TESTS:-tests} -x -q -m gpu 2>&1 | tail -3
for WL in "$@"; do
python bench.py --workload $WL --cpu-samples 0 --steps 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$WL', d['value'], 'Msps', d['ms_per_step'], 'ms', d['check'], d['fec_stats'], d['demod_stats'])
print('  ', ' '.join(f\"{k.replace('k_','')}={v['ms_per_step']}/{v['launches_per_step']:.0f}\" for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:14]))"
done
