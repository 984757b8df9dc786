#!/bin/bash
# Round 4, visit H: the Meteor LRPT decoder on the GPU -- parity tests, the plugin module through the minihost, its rate
TAG=${1:-r04_h}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lrpt_gpu.py -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest_lrpt.txt
timeout 600 python -m pytest tests/test_plugin_minihost_gpu.py -m gpu -q -x -k "lrpt" 2>&1 | tail -8 | tee $OUT/pytest_plugin.txt
timeout 600 python tools/bench_lrpt.py 2>&1 | tail -3 | tee $OUT/bench_lrpt.json
timeout 600 python tools/bench_lrpt.py --frames 32768 --cpu-frames 0 2>&1 | tail -1 | tee $OUT/bench_lrpt_32k.json
