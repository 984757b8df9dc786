#!/bin/bash
# Round 3, visit F: the GPU tests written since visit E (ndsp chain + its plugin block, punctured short CADUs, minihost fixes, DVB-S2 descrambler /
# de-interleaver, viterbi27, long Viterbi segments, freq_shift), the ndsp chain at bench size, the M&M warm-up / lane A/B on MetOp.
TAG=${1:-r03_f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ndsp_gpu.py tests/test_plugin_minihost_gpu.py tests/test_zz_punctured_gpu.py tests/test_dvbs2_gpu.py tests/test_zy_demod_additions_gpu.py -m gpu -q --durations=5 -k "ndsp or minihost or short_cadus or deinterleave or descrambl or freq_shift or chain" 2>&1 | tail -25 > $OUT/pytest_new.txt; tail -14 $OUT/pytest_new.txt
timeout 300 python -m pytest tests/test_fec_gpu.py -m gpu -q -k "viterbi27 or long_segments" 2>&1 | tail -4 | tee $OUT/pytest_fec_new.txt
timeout 400 python tools/bench_ndsp.py > $OUT/bench_ndsp.json 2> $OUT/bench_ndsp.err || { echo "bench_ndsp rc $?"; tail -15 $OUT/bench_ndsp.err; }
head -c 2500 $OUT/bench_ndsp.json; echo
timeout 400 python tools/bench_ndsp.py --constellation bpsk --samples 268435456 > $OUT/bench_ndsp_bpsk.json 2> $OUT/bench_ndsp_bpsk.err || { echo "bench_ndsp bpsk rc $?"; tail -8 $OUT/bench_ndsp_bpsk.err; }
head -c 1500 $OUT/bench_ndsp_bpsk.json; echo
timeout 600 python tools/ab_demod.py --workload metop_ahrpt "" "SDHIP_W_MM=5632" "SDHIP_W_MM=5632,SDHIP_LANES_MM=130560" "SDHIP_W_MM=7680,SDHIP_LANES_MM=98304" > $OUT/ab_metop.txt 2> $OUT/ab_metop.err; cat $OUT/ab_metop.txt; tail -3 $OUT/ab_metop.err
