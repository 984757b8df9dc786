#!/bin/bash
# Round 5, visit L (kernel sources untouched): the clock recovery storing the module's int8 soft symbols itself (SDHIP_MM_Q8=1, round 2's experiment: slower then,
# when k_mm was bound by its instruction issue) against the float rows + k_quantize, on today's k_mm
TAG=${1:-r05_l}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/ab_demod.py --workload metop_ahrpt --steps 4 --warmup 2 "" "SDHIP_MM_Q8=1" "" "SDHIP_MM_Q8=1" 2> $OUT/ab_metop_ahrpt.err | tee $OUT/ab_metop_ahrpt.txt
