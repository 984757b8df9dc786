#!/bin/bash
# Round 5, visit D: k_mm with cooperative loads: bit-identity test + demod / ndsp / dvbs2 suites on the device, A/B against SDHIP_COOP=0, lanes
TAG=${1:-r05_d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zy_demod_additions_gpu.py tests/test_demod_gpu.py tests/test_ndsp_gpu.py -m gpu -q -x -k "not full_size and not margin_sweep and not three_passes" 2>&1 | tail -6 | tee $OUT/pytest_sel.txt
timeout 300 python tools/ab_demod.py --workload metop_ahrpt --steps 4 --warmup 2 "" "SDHIP_COOP=0" "SDHIP_LANES_MM=130560" "SDHIP_LANES_MM=65280" 2> $OUT/ab_metop_ahrpt.err | tee $OUT/ab_metop_ahrpt.txt
timeout 300 python tools/ab_demod.py --workload npp_hrd --steps 4 --warmup 2 "" "SDHIP_LANES_MM=130560" 2> $OUT/ab_npp_hrd.err | tee $OUT/ab_npp_hrd.txt
timeout 300 python tools/ab_demod.py --workload goes_hrit --steps 4 --warmup 2 "" "SDHIP_COOP=0" "SDHIP_LANES_MM=98304" 2> $OUT/ab_goes_hrit.err | tee $OUT/ab_goes_hrit.txt
