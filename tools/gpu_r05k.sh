#!/bin/bash
# Round 5, visit K (kernel sources untouched since the closing visit): the plugin's ZIQ reader on the device -- the plugin's whole minihost suite (the new
# test_ziq_container_through_the_plugin and the DVB-S2 module's ziq leg among them)
TAG=${1:-r05_k}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_plugin_minihost_gpu.py -m gpu -q --durations=8 2>&1 | tail -25 | tee $OUT/pytest_plugin.txt
