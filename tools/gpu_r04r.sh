#!/bin/bash
# Round 4, visit R: how often k_mm lands in its fast mode, per dynamic-LDS pad (fresh handles each time, one process per pad so that the order of
# the configurations cannot matter)
TAG=${1:-r04_r}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
W="--workload metop_ahrpt --steps 2 --warmup 1 --cpu-samples 0"
for pad in 0 2048 4096 0 2048; do
  c="SDHIP_MM_LDS_PAD=$pad"
  timeout 900 python tools/ab_demod.py $W "$c" "$c" "$c" "$c" "$c" "$c" 2>&1 | grep -o '"cfg": "[^"]*", "ms_per_step": [0-9.]*\|"k_mm": [0-9.]*' | paste - - | tee -a $OUT/mm_modes.txt
done
