#!/bin/bash
# Round 5, visit Q (kernel sources untouched): ccsds_simple_psk_decoder with hard_symbols input through the plugin on the device
TAG=${1:-r05_q}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_plugin_minihost_gpu.py -m gpu -q -k "hard_symbols or uncovered" 2>&1 | tail -6 | tee $OUT/pytest_hard_symbols.txt
