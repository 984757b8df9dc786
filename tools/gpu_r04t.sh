#!/bin/bash
# Round 4, visit T: k_mm's mode against the addresses of its buffers (SDHIP_PRINT_ADDR), fresh handles on freshly allocated blocks
TAG=${1:-r04_t}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
W="--workload metop_ahrpt --steps 2 --warmup 0 --cpu-samples 0"
for rep in 1 2; do
  SDHIP_PRINT_ADDR=1 timeout 900 python tools/ab_demod.py $W "" "" "" "" "" "" "" "" 2>&1 | grep -o 'mm buffers: in 0x[0-9a-f]*  symbols 0x[0-9a-f]*\|"k_mm": [0-9.]*' | uniq | tee -a $OUT/mm_addr.txt
done
