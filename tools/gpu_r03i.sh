#!/bin/bash
# Round 3, visit I: M&M lane-count scan (chunk length = samples / lanes) on MetOp and NPP
TAG=${1:-r03_i}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for WL in metop_ahrpt npp_hrd; do
timeout 1500 python tools/ab_demod.py --workload $WL "" "SDHIP_LANES_MM=73728" "SDHIP_LANES_MM=81920" "SDHIP_LANES_MM=90112" "SDHIP_LANES_MM=98304" "SDHIP_LANES_MM=100352" "SDHIP_LANES_MM=106496" "SDHIP_LANES_MM=114688" "SDHIP_LANES_MM=122880" "SDHIP_LANES_MM=130560" \
  > $OUT/ab_$WL.txt 2> $OUT/ab_$WL.err; tail -2 $OUT/ab_$WL.err
python - <<PY
import json
print("$WL")
for l in open("$OUT/ab_$WL.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(f'{d["cfg"]:28s} step {d["ms_per_step"]:7.3f}  k_mm {d["kernels_ms"]["k_mm"]:7.3f} k_afc {d["kernels_ms"]["k_afc"]:7.3f} within1e-5 {d["parity"]["frac_within_1e-5"]:.6f}  int8eq {d["parity"]["frac_int8_equal"]:.6f} fixed {d["first_pass"]["fixed"]} ident {d["parity"]["cadus_identical"]}')
PY
done
