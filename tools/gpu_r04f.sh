#!/bin/bash
# Round 4, visit F: the whole GPU suite (mid-round check)
TAG=${1:-r04_f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -m gpu -q 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt
