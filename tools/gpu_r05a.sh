#!/bin/bash
# Round 5, visit A: the three prepared kernel branches on the device (k_rs_screen4, k_mm ring mirror, k_vit2_prep<MODE, PHASE>): the GPU suites that exercise them
# (full-size reference decodes and margin sweeps left to the closing visit), A/B on the three workloads against the old kernels where a switch exists,
# k_mm's launch-time mode with and without the allocation spacer over several processes
TAG=${1:-r05_a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fec_gpu.py tests/test_golden_gpu.py tests/test_zz_punctured_gpu.py tests/test_fy3_gpu.py tests/test_lrpt_gpu.py tests/test_demod_gpu.py tests/test_ndsp_gpu.py -m gpu -q -x -k "not full_size and not margin_sweep" 2>&1 | tail -8 | tee $OUT/pytest_sel.txt
OLD="SDHIP_RS_SCREEN4=0,SDHIP_VIT2_PREP_TEMPL=0"
for WL in metop_ahrpt npp_hrd goes_hrit; do
  timeout 400 python tools/ab_demod.py --workload $WL --steps 4 --warmup 2 "" "$OLD" "SDHIP_ALLOC_SPACER=1" 2> $OUT/ab_$WL.err | tee $OUT/ab_$WL.txt
done
for rep in 1 2 3; do
  timeout 200 python tools/ab_demod.py --workload metop_ahrpt --steps 2 --warmup 0 --cpu-samples 0 "" "SDHIP_ALLOC_SPACER=1" "" "SDHIP_ALLOC_SPACER=1" 2>> $OUT/modes.err | grep -o '"cfg": "[^"]*"\|"k_mm": [0-9.]*' | tr '\n' ' ' | tee -a $OUT/mm_modes.txt
  echo | tee -a $OUT/mm_modes.txt
done
