#!/bin/bash
# Round 4, visit S: do k_mm's two modes follow the memory a handle gets? Fresh handles on recycled blocks (--pool 1: same device memory every time) against
# fresh handles on freshly allocated blocks
TAG=${1:-r04_s}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
W="--workload metop_ahrpt --steps 2 --warmup 1 --cpu-samples 0"
for rep in 1 2 3; do
  echo "pool, process $rep" | tee -a $OUT/mm_modes.txt
  timeout 900 python tools/ab_demod.py $W --pool 1 "" "" "" "" "" "" 2>&1 | grep -o '"k_mm": [0-9.]*' | tr '\n' ' ' | tee -a $OUT/mm_modes.txt; echo | tee -a $OUT/mm_modes.txt
done
