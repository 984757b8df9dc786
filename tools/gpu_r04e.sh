#!/bin/bash
# Round 4, visit E: cheap A/Bs on the driver workload -- M&M interpolator arm stride (LDS bank groups), Costas warm-up in the fused stage, modules overlapped
TAG=${1:-r04_e}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python tools/ab_demod.py --workload metop_ahrpt --steps 4 --warmup 2 "" "SDHIP_MM_ARM_STRIDE=12" "SDHIP_COSTAS_TAUS=14" "SDHIP_COSTAS_TAUS=14,SDHIP_MM_ARM_STRIDE=12" "" 2>&1 | tail -40 | tee $OUT/ab_metop.txt
echo "== modules overlapped (--pipeline)"
timeout 600 python bench.py --steps 6 --warmup 2 --cpu-samples 0 --others 0 --next-rows 0 --pipeline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline', d['value'], d['ms_per_step'])" | tee $OUT/pipeline.txt
timeout 600 python bench.py --steps 6 --warmup 2 --cpu-samples 0 --others 0 --next-rows 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', d['value'], d['ms_per_step'])" | tee -a $OUT/pipeline.txt
