#!/bin/bash
# Round 5, visit E: the tests written since visit D on the device (fengyun_mpt_decoder, LRPT m2x through the plugin, DVB-S2 freq_prop hand-over, cooperative
# lanes incl. the clock recovery's), then the clock recovery at its new default of 65 280 lanes: cooperative loads on / off, shorter warm-ups
TAG=${1:-r05_e}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fy3_gpu.py tests/test_plugin_minihost_gpu.py tests/test_dvbs2_gpu.py tests/test_zy_demod_additions_gpu.py -m gpu -q -x -k "mpt or lrpt or freq_prop or cooperative or fy3_module" 2>&1 | tail -6 | tee $OUT/pytest_new.txt
timeout 300 python tools/ab_demod.py --workload metop_ahrpt --steps 4 --warmup 2 "" "SDHIP_COOP_MM=1" "SDHIP_W_MM=10752" "SDHIP_W_MM=8192" "SDHIP_W_MM=10752,SDHIP_COOP_MM=1" 2> $OUT/ab_metop_ahrpt.err | tee $OUT/ab_metop_ahrpt.txt
timeout 300 python tools/ab_demod.py --workload npp_hrd --steps 4 --warmup 2 "" "SDHIP_W_MM=10752" 2> $OUT/ab_npp_hrd.err | tee $OUT/ab_npp_hrd.txt
timeout 300 python tools/ab_demod.py --workload goes_hrit --steps 4 --warmup 2 "" "SDHIP_COOP_MM=1" 2> $OUT/ab_goes_hrit.err | tee $OUT/ab_goes_hrit.txt
