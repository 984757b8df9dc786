#!/bin/bash
# Round 3, visit J: chunk-length residue scan -- the lane stride (chunk length x 8 bytes) against the HBM channel interleave
TAG=${1:-r03_j}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
WL=${2:-metop_ahrpt}
timeout 1500 python tools/ab_demod.py --workload $WL "" "SDHIP_CHUNK_MM=21632" "SDHIP_CHUNK_MM=21696" "SDHIP_CHUNK_MM=21760" "SDHIP_CHUNK_MM=21824" "SDHIP_CHUNK_MM=21888" "SDHIP_CHUNK_MM=22016" "SDHIP_CHUNK_MM=22144" \
  "SDHIP_CHUNK_MM=32896" "SDHIP_CHUNK_MM=33024" "SDHIP_CHUNK_MM=33152" \
  "SDHIP_CHUNK_COSTAS=16448" "SDHIP_CHUNK_COSTAS=16576" "SDHIP_CHUNK_COSTAS=16640" "SDHIP_CHUNK_COSTAS=16768" "SDHIP_CHUNK_COSTAS=16384" \
  > $OUT/ab_$WL.txt 2> $OUT/ab_$WL.err; tail -2 $OUT/ab_$WL.err
python - <<PY
import json
print("$WL")
for l in open("$OUT/ab_$WL.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(f'{d["cfg"]:28s} step {d["ms_per_step"]:7.3f}  k_mm {d["kernels_ms"]["k_mm"]:7.3f} k_afc {d["kernels_ms"]["k_afc"]:7.3f} within1e-5 {d["parity"]["frac_within_1e-5"]:.6f} fixed {d["first_pass"]["fixed"]} ident {d["parity"]["cadus_identical"]}')
PY
