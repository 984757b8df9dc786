#!/bin/bash
# First GPU visit of the next round (≈4-5 GPU-minutes; GOES benches take ≈8 s each, every metop/npp bench ≈40-50 s because the
# 17 GB stream is synthesised per process -- the round-1 sweep ran out of budget on exactly that):
#   1. full parity suite + GOES bench/rocprof/PMC round (tools/gpu_round.sh)            ≈1 min
#   2. A/B of the experimental checkpointed re-run exit (SDHIP_MM_CKPT=1): demod tests + bench ≈30 s
#   3. lane-target sweep on GOES (cheap)                                                 ≈40 s
#   4. one npp_hrd and one metop_ahrpt line with the shipped defaults (doc table)        ≈100 s
# Usage: tools/gpu_next.sh <tag>
TAG=${1:-r02_a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
NO_SQ=1 tools/gpu_round.sh $TAG goes_hrit
echo "== SDHIP_CKPT=1: demod parity tests, then bench" | tee $OUT/ckpt.txt
SDHIP_CKPT=1 python -m pytest tests/test_demod_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee -a $OUT/ckpt.txt
tools/sweep.sh SDHIP_CKPT=0 SDHIP_CKPT=1 SDHIP_MM_SPLIT=1 2>&1 | tee -a $OUT/ckpt.txt
tools/sweep.sh SDHIP_LANES_COSTAS=130560 SDHIP_LANES_COSTAS=163840 SDHIP_LANES_MM=61440 "SDHIP_LANES_MM=57344 SDHIP_LANES_AGC=49152" 2>&1 | tee $OUT/lanes_goes.txt
tools/sweep_wl.sh npp_hrd A=1 2>&1 | tee $OUT/npp.txt
tools/sweep_wl.sh metop_ahrpt A=1 2>&1 | tee $OUT/metop.txt
