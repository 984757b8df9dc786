#!/bin/bash
# Round 4, visit Q: k_vit2h_acs at 3 and 4 waves per SIMD (variant libraries), k_mm with dynamic LDS padding (6 blocks per CU instead of 7), repeated
# fresh handles to see both of its modes
TAG=${1:-r04_q}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
W="--workload metop_ahrpt --steps 3 --warmup 1 --cpu-samples 0"
for v in w3 w4; do
  SDHIP_LIB=$PWD/satdump_amd/lib/libsdhip_$v.so timeout 600 python tools/ab_demod.py $W "" 2>&1 | tail -1 | cut -c1-700 | tee $OUT/ab_$v.txt
done
timeout 1200 python tools/ab_demod.py $W "" "SDHIP_MM_LDS_PAD=2048" "" "SDHIP_MM_LDS_PAD=2048" "" "SDHIP_MM_LDS_PAD=2048" "SDHIP_MM_LDS_PAD=4096" "SDHIP_MM_LDS_PAD=4096" 2>&1 | tail -8 | cut -c1-600 | tee $OUT/ab_mm_pad.txt
