#!/bin/bash
# Round 4, visit P: the lane-per-segment Viterbi decoder with path histories (k_vit2h_acs / k_vit2h_tb): the FEC suites, then A/B against the per-step
# decision words (SDHIP_VIT2_HIST=0) on the three driver workloads and the FengYun decoder
TAG=${1:-r04_p}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fec_gpu.py tests/test_golden_gpu.py tests/test_zz_punctured_gpu.py tests/test_fy3_gpu.py tests/test_lrpt_gpu.py -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest_fec.txt
timeout 1200 python tools/ab_demod.py --workload metop_ahrpt --steps 4 --warmup 2 "" "SDHIP_VIT2_HIST=0" "" 2>&1 | tail -3 | tee $OUT/ab_metop.txt
timeout 600 python tools/ab_demod.py --workload goes_hrit --steps 4 --warmup 2 "" "SDHIP_VIT2_HIST=0" 2>&1 | tail -2 | tee $OUT/ab_goes.txt
timeout 600 python tools/ab_demod.py --workload npp_hrd --steps 4 --warmup 2 "" "SDHIP_VIT2_HIST=0" 2>&1 | tail -2 | tee $OUT/ab_npp.txt
timeout 600 python tools/bench_fy3.py --cpu-frames 0 > $OUT/bench_fy3.json 2> $OUT/bench_fy3.err; cat $OUT/bench_fy3.json
