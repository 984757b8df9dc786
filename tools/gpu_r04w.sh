#!/bin/bash
# Round 4, visit W (scouting for round 5, product untouched): k_mm's two launch-time modes against WHERE the handle's large buffers land -- fresh processes with
# tools/ubench/malloc_shim.cpp preloaded (spacer in front of / doubled engine allocations), three fresh handles per process
TAG=${1:-r04_w}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
W="--workload metop_ahrpt --steps 2 --warmup 0 --cpu-samples 0"
SHIM=$PWD/tools/ubench/libmalloc_shim.so
for rep in ${REPS:-1}; do
  for m in ${MODES:-"0" "3" "2"}; do
    set -- $m
    echo "mode $1 ${2:+spacer $2 MiB} (process $rep)" | tee -a $OUT/mm_shim.txt
    LD_PRELOAD=$SHIM SHIM_MODE=$1 SHIM_MB=${2:-0} timeout 200 python tools/ab_demod.py $W "" "" 2> $OUT/err_$1_${2:-0}_$rep.txt | grep -o '"k_mm": [0-9.]*' | tr '\n' ' ' | tee -a $OUT/mm_shim.txt
    echo | tee -a $OUT/mm_shim.txt
    grep -c "shim" $OUT/err_$1_${2:-0}_$rep.txt | tee -a $OUT/mm_shim.txt
  done
done
grep -h "shim" $OUT/err_3_0_1.txt | head -12
