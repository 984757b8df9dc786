#!/bin/bash
# Round 5, visit O (final sources, hash as the second closing visit's): the three full-size reference decodes the second closing visit left out of its suite run
TAG=${1:-r05_o}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
SDHIP_FINAL=1 timeout 600 python -m pytest tests/ -m gpu -q --durations=5 -k "full_size" 2>&1 | tail -12 | tee $OUT/pytest_full_size.txt
