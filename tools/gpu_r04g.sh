#!/bin/bash
# Round 4, visit G: the packed ACS step with high-byte metrics / tag decisions (k_vit2_acs) and the Gardner lanes: parity tests, then the driver workload's kernel times
TAG=${1:-r04_g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fec_gpu.py tests/test_golden_gpu.py tests/test_zz_punctured_gpu.py tests/test_ndsp_gpu.py -m gpu -q -x 2>&1 | tail -15 | tee $OUT/pytest_fec.txt
timeout 900 python -m pytest tests/test_plugin_minihost_gpu.py -m gpu -q -x -k "ndsp_single or metop" 2>&1 | tail -8 | tee $OUT/pytest_plugin.txt
timeout 1200 python tools/ab_demod.py --workload metop_ahrpt --steps 4 --warmup 2 "" "" 2>&1 | tail -12 | tee $OUT/ab_metop.txt
timeout 600 python tools/ab_demod.py --workload goes_hrit --steps 4 --warmup 2 "" 2>&1 | tail -6 | tee $OUT/ab_goes.txt
timeout 600 python tools/ab_demod.py --workload npp_hrd --steps 4 --warmup 2 "" 2>&1 | tail -6 | tee $OUT/ab_npp.txt
