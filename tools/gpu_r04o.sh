#!/bin/bash
# Round 4, visit O: fengyun_ahrpt_decoder (SDHIP_DEC_FENGYUN_AHRPT) on the GPU -- its parity tests, the plugin module through the minihost, the FEC / golden
# suites behind the decode-run refactor (vit_run), the decoder's rate
TAG=${1:-r04_o}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fy3_gpu.py -m gpu -q -x --durations=5 2>&1 | tail -15 | tee $OUT/pytest_fy3.txt
timeout 600 python -m pytest tests/test_plugin_minihost_gpu.py -m gpu -q -x -k "fy3 or lrpt or boundary" 2>&1 | tail -6 | tee $OUT/pytest_plugin.txt
timeout 900 python -m pytest tests/test_fec_gpu.py tests/test_golden_gpu.py -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_fec.txt
timeout 600 python tools/bench_fy3.py > $OUT/bench_fy3.json 2> $OUT/bench_fy3.err || tail -5 $OUT/bench_fy3.err
cat $OUT/bench_fy3.json
SDHIP_FEC_BATCH=8192 timeout 600 python tools/bench_fy3.py --cpu-frames 0 > $OUT/bench_fy3_b8192.json 2>> $OUT/bench_fy3.err; cat $OUT/bench_fy3_b8192.json
