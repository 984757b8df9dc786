#!/bin/bash
# Round 3, visit G: the ndsp chain with the chunk-parallel arithmetic in its stand-alone AGC stage (A/B of the load-group depth)
TAG=${1:-r03_g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ndsp_gpu.py -m gpu -q 2>&1 | tail -3
for e in "" "SDHIP_AGC_DEPTH=8" "SDHIP_AGC_TAUS=10" "SDHIP_AGC_TAUS=24"; do
  echo "== $e"; env $e timeout 400 python tools/bench_ndsp.py --cpu-samples 12000000 > $OUT/bench_ndsp_${e//[=]/_}.json 2> $OUT/bench_ndsp.err || { echo "rc $?"; tail -8 $OUT/bench_ndsp.err; }
  python - <<PY
import json
d=json.load(open("$OUT/bench_ndsp_${e//[=]/_}.json")); print(d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["parity_vs_reference_first_call"]["frac_within_1e5"], d["parity_vs_reference_first_call"]["median_rel"], d["steady_chunks"])
PY
done
