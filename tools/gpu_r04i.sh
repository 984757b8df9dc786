#!/bin/bash
# Round 4, visit I: where a k_ldpc_trial workgroup spends its time (SDHIP_LDPC_PROBE)
TAG=${1:-r04_i}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
SDHIP_LDPC_PROBE=1 timeout 600 python tools/bench_dvbs2.py --rate 2/3 --sigma 13 --front 0 --cpu-frames 0 --sync-frames 0 2>&1 | tail -12 | cut -c1-1500 | tee $OUT/probe_2_3.txt
timeout 600 python tools/bench_dvbs2.py --rate 2/3 --sigma 13 --front 0 --cpu-frames 0 --sync-frames 0 2>&1 | tail -1 | cut -c1-1200 | tee $OUT/plain_2_3.txt
