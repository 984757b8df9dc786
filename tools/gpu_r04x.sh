#!/bin/bash
# Round 4, visit X (after the closing visit, same kernel sources): the plugin's wav / RF64 reader through the minihost on the device, and the SQ counter
# passes of visit Y for the other two workloads
TAG=${1:-r04_x}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_plugin_minihost_gpu.py -m gpu -q -k "wav or stock_ids" 2>&1 | tail -5 | tee $OUT/pytest_wav.txt
for WL in npp_hrd goes_hrit; do
  bash tools/gpu_r04y.sh $TAG $WL > $OUT/sq_$WL.txt 2>&1
  head -6 $OUT/${WL}_sq.csv | cut -c1-250
done
