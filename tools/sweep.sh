#!/bin/bash
# tuning sweep on the GPU box: bench the goes workload under several chunk / warm-up settings (experiments only)
run() { echo "== $*"; env SDHIP_DEBUG=1 "$@" python bench.py --steps 2 --warmup 1 --cpu-samples 0 2>gpurun_out/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('Msps',d['value'],'ms',d['ms_per_step'],'check',d['check'],'stats',d['demod_stats'])
print('  '+' '.join(f\"{n.replace('k_chunks<','').replace('Stage>','')}={v['ms_per_step']}/{v['launches_per_step']}\" for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:8]))
"; grep "re-run" gpurun_out/sweep.err | tail -3;  grep -m2 "rejected" gpurun_out/sweep.err; }
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run A=1
run SDHIP_W_MM=8192
run SDHIP_W_MM=4096 SDHIP_MM_TOL_MILLI=100
run SDHIP_W_MM=8192 SDHIP_MM_TOL_MILLI=100
run SDHIP_CHUNK=3584
run SDHIP_CHUNK=1792
