#!/bin/bash
# Round 5, visit C: k_afc with cooperative (coalesced) access to its 64 streams per wave: the new bit-identity test + demod suites, A/B against SDHIP_COOP=0
TAG=${1:-r05_c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zy_demod_additions_gpu.py tests/test_demod_gpu.py -m gpu -q -x -k "cooperative or chunked_mode or exact_mode_bit or arm_flip" 2>&1 | tail -8 | tee $OUT/pytest_sel.txt
for WL in metop_ahrpt npp_hrd goes_hrit; do
  timeout 300 python tools/ab_demod.py --workload $WL --steps 4 --warmup 2 "" "SDHIP_COOP=0" 2> $OUT/ab_$WL.err | tee $OUT/ab_$WL.txt
done
