#!/bin/bash
# Round 4, visit L: k_ldpc_trial -- wide layers fetched one ahead, link inputs kept in registers, parity check four layers at a time
TAG=${1:-r04_l}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dvbs2_gpu.py -m gpu -q -x -k "ldpc or code or batch" 2>&1 | tail -4 | tee $OUT/pytest_dvbs2.txt
SDHIP_LDPC_PROBE=1 timeout 600 python tools/bench_dvbs2.py --rate 2/3 --sigma 13 --front 0 --cpu-frames 0 --sync-frames 0 2>&1 | tail -2 | cut -c1-1500 | tee $OUT/probe_2_3.txt
for r in 1/2 3/4 9/10; do timeout 600 python tools/bench_dvbs2.py --rate $r --front 0 --cpu-frames 0 --sync-frames 0 2>&1 | tail -1 | cut -c1-500 | tee -a $OUT/rates.txt; done
timeout 900 python tools/bench_dvbs2_demod.py --cpu-frames 0 2>&1 | tail -1 | cut -c1-400 | tee $OUT/demod_8psk.txt
