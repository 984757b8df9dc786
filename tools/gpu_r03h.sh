#!/bin/bash
# Round 3, visit H: M&M warm-up x lanes grid on MetOp (step time, k_mm, soft parity of the first pass)
TAG=${1:-r03_h}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python tools/ab_demod.py --workload metop_ahrpt "" \
  "SDHIP_W_MM=8192" "SDHIP_W_MM=7168" "SDHIP_W_MM=6144" \
  "SDHIP_LANES_MM=98304" "SDHIP_W_MM=9216,SDHIP_LANES_MM=98304" "SDHIP_W_MM=8192,SDHIP_LANES_MM=98304" "SDHIP_W_MM=7168,SDHIP_LANES_MM=98304" \
  "SDHIP_LANES_MM=130560" "SDHIP_W_MM=9216,SDHIP_LANES_MM=130560" "SDHIP_W_MM=8192,SDHIP_LANES_MM=130560" "SDHIP_W_MM=7168,SDHIP_LANES_MM=130560" \
  > $OUT/ab_metop.txt 2> $OUT/ab_metop.err; tail -3 $OUT/ab_metop.err
python - <<PY
import json
for l in open("$OUT/ab_metop.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(f'{d["cfg"]:45s} step {d["ms_per_step"]:7.3f}  k_mm {d["kernels_ms"]["k_mm"]:7.3f}  within1e-5 {d["parity"]["frac_within_1e-5"]:.6f}  int8eq {d["parity"]["frac_int8_equal"]:.6f} fixed {d["first_pass"]["fixed"]} cadus_identical {d["parity"]["cadus_identical"]}')
PY
