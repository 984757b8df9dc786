#!/bin/bash
# Round 5, visit B: k_mm with the 16-slot ring (two waves per SIMD fit) and the wave-uniform fast paths: demod / ndsp / dvbs2 GPU suites, then lanes A/B on the three workloads
TAG=${1:-r05_b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_demod_gpu.py tests/test_zy_demod_additions_gpu.py tests/test_ndsp_gpu.py tests/test_golden_gpu.py -m gpu -q -x -k "not full_size_metop and not full_size_npp and not margin_sweep" 2>&1 | tail -8 | tee $OUT/pytest_sel.txt
timeout 300 python tools/ab_demod.py --workload metop_ahrpt --steps 4 --warmup 2 "" "SDHIP_LANES_MM=130560" "SDHIP_LANES_MM=163840" "SDHIP_LANES_MM=130560,SDHIP_W_MM=6144" 2> $OUT/ab_metop.err | tee $OUT/ab_metop_ahrpt.txt
timeout 300 python tools/ab_demod.py --workload npp_hrd --steps 4 --warmup 2 "" "SDHIP_LANES_MM=130560" 2> $OUT/ab_npp.err | tee $OUT/ab_npp_hrd.txt
timeout 300 python tools/ab_demod.py --workload goes_hrit --steps 4 --warmup 2 "" "SDHIP_LANES_MM=98304" "SDHIP_LANES_MM=130560" 2> $OUT/ab_goes.err | tee $OUT/ab_goes_hrit.txt
