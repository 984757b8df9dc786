#!/bin/bash
# Round 3, visit K: k_mm / k_afc time split into the main launch and the re-run launches of the certificate loop
TAG=${1:-r03_k}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
WL=${2:-metop_ahrpt}
SDHIP_PRINT_ADDR=1 timeout 1500 python tools/ab_demod.py --workload $WL "" "" "SDHIP_VIT2_SEG=4096" "SDHIP_VIT2_SEG=1024" "" "SDHIP_COSTAS_TAUS=18" "" \
  > $OUT/ab_$WL.txt 2> $OUT/ab_$WL.err; grep "mm buffers" $OUT/ab_$WL.err | uniq -c
python - <<PY
import json
print("$WL")
for l in open("$OUT/ab_$WL.txt"):
    if l.startswith("{"):
        d=json.loads(l); k=d["kernels_ms"]
        print(f'{d["cfg"]:28s} step {d["ms_per_step"]:7.3f}  k_mm {k["k_mm"]:7.3f} (re-run {k.get("k_mm (re-run launches, included in k_mm)",0):6.3f}) k_afc {k["k_afc"]:7.3f} (re-run {k.get("k_afc (re-run launches, included in k_afc)",0):6.3f}) within1e-5 {d["parity"]["frac_within_1e-5"]:.6f} steady fixed {d["steady"]["fixed"]} ident {d["parity"]["cadus_identical"]} | ' + ' '.join(f"{a.replace('k_','')}={b}" for a,b in list(k.items())[2:9]))
PY
