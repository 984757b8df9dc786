#!/bin/bash
# MetOp sweep: bytes per lane per load group and lanes of the AGC / Costas stages
TAG=${1:-r02_j}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --cpu-samples 0 --steps 3 --warmup 1 > $OUT/$name.json 2> $OUT/$name.err || { echo FAILED $name; tail -5 $OUT/$name.err; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json"))
k=d['kernels']
print("%-28s %8.1f ms  agc %.2f  costas %.2f  mm %.2f  fir %.2f  ok=%s" % ("$name", d["ms_per_step"], k['k_chunks<AgcStage>']['ms_per_step'], k['k_chunks<CostasStage>']['ms_per_step'], k['k_mm']['ms_per_step'], k['k_fir_window']['ms_per_step'], d["check"]["cadus_matching_transmitted"]))
PY
}
run base X=1
run agc_d8 SDHIP_AGC_DEPTH=8
run agc_d2 SDHIP_AGC_DEPTH=2
run agc_l130k SDHIP_LANES_AGC=130560
run agc_d2_l130k SDHIP_AGC_DEPTH=2 SDHIP_LANES_AGC=130560
run cos_l261k SDHIP_LANES_COSTAS=261120
run cos_l130k SDHIP_LANES_COSTAS=130560
run cos_d4_l130k SDHIP_COSTAS_DEPTH=4 SDHIP_LANES_COSTAS=130560
run cos_d8_l65k SDHIP_COSTAS_DEPTH=8 SDHIP_LANES_COSTAS=65280
run cos_d4_l65k SDHIP_COSTAS_DEPTH=4 SDHIP_LANES_COSTAS=65280
