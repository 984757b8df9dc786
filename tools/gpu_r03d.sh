#!/bin/bash
# Round 3, visit D: the whole GPU suite (incl. the two 17 GB reference decodes), Viterbi segment length A/B
TAG=${1:-r03_d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 > $OUT/pytest_gpu.txt; tail -14 $OUT/pytest_gpu.txt
timeout 600 python tools/ab_demod.py --workload metop_ahrpt --cpu-samples 0 "" "SDHIP_VIT2_SEG=2048" "SDHIP_VIT2_SEG=512" > $OUT/ab_metop.txt 2> $OUT/ab_metop.err; cat $OUT/ab_metop.txt; tail -3 $OUT/ab_metop.err
timeout 300 python tools/ab_demod.py --workload npp_hrd --cpu-samples 0 "" "SDHIP_VIT2_SEG=2048" > $OUT/ab_npp.txt 2> $OUT/ab_npp.err; cat $OUT/ab_npp.txt
