#!/bin/bash
TAG=${1:-r02_p}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --cpu-samples 0 --steps 3 --warmup 1 > $OUT/$name.json 2> $OUT/$name.err || { echo FAILED $name; tail -3 $OUT/$name.err; return; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json"))
k=d['kernels']
g=lambda n: k.get(n,{}).get('ms_per_step',0)
print("%-22s %8.1f ms  agcfir %.2f costas %.2f  mm %.2f  ok=%s" % ("$name", d["ms_per_step"], g('k_chunks<AgcFirStage>'), g('k_chunks<CostasStage>'), g('k_mm'), d["check"]["cadus_matching_transmitted"]))
PY
}
run base X=1
run cos_d4_l65k SDHIP_COSTAS_DEPTH=4 SDHIP_LANES_COSTAS=65280
run cos_d8_l65k SDHIP_COSTAS_DEPTH=8 SDHIP_LANES_COSTAS=65280
run cos_d4_l98k SDHIP_COSTAS_DEPTH=4 SDHIP_LANES_COSTAS=97920
run cos_d4_l130k SDHIP_COSTAS_DEPTH=4 SDHIP_LANES_COSTAS=130560
