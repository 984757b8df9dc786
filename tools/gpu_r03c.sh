#!/bin/bash
# Round 3, visit C: fast arithmetic of the chunk-parallel mode (A/B), DVB-S2 LDPC + BCH parity on the GPU and a first throughput line.
TAG=${1:-r03_c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dvbs2_gpu.py -m gpu -q 2>&1 | tail -8 > $OUT/pytest_dvbs2.txt; tail -4 $OUT/pytest_dvbs2.txt
timeout 900 python -m pytest tests/test_demod_gpu.py tests/test_golden_gpu.py -m gpu -q -k "margin or chunked or golden or exact_mode_bit" 2>&1 | tail -6 > $OUT/pytest_demod.txt; tail -3 $OUT/pytest_demod.txt
timeout 600 python tools/ab_demod.py --workload metop_ahrpt "" "SDHIP_FAST_MATH=0" > $OUT/ab_metop.txt 2> $OUT/ab_metop.err; cat $OUT/ab_metop.txt; tail -3 $OUT/ab_metop.err
timeout 300 python tools/ab_demod.py --workload goes_hrit "" "SDHIP_FAST_MATH=0" > $OUT/ab_goes.txt 2> $OUT/ab_goes.err; cat $OUT/ab_goes.txt
timeout 300 python tools/ab_demod.py --workload npp_hrd "" "SDHIP_FAST_MATH=0" > $OUT/ab_npp.txt 2> $OUT/ab_npp.err; cat $OUT/ab_npp.txt
for r in "2/3" "9/10" "1/2"; do timeout 300 python tools/bench_dvbs2.py --rate $r --sigma $( [ $r = "2/3" ] && echo 13 || ([ $r = "9/10" ] && echo 7.3 || echo 17.5) ) >> $OUT/bench_dvbs2.txt 2>> $OUT/bench_dvbs2.err; done; cat $OUT/bench_dvbs2.txt; tail -3 $OUT/bench_dvbs2.err
