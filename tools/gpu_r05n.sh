#!/bin/bash
# Round 5, visit N: visit M's A/B once more on GOES (BPSK: one sample there had the int8 rows 0.4 ms behind the float rows -- the kernel, or k_mm's two launch-time
# modes?) and on NPP
TAG=${1:-r05_n}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python tools/ab_demod.py --workload goes_hrit --steps 4 --warmup 2 --cpu-samples 2000000 "SDHIP_MM_Q8=0" "" "SDHIP_MM_Q8=0" "" 2> $OUT/ab_goes_hrit.err | tee $OUT/ab_goes_hrit.txt | cut -c1-100
timeout 200 python tools/ab_demod.py --workload npp_hrd --steps 4 --warmup 2 --cpu-samples 2000000 "" "SDHIP_MM_Q8=0" 2> $OUT/ab_npp_hrd.err | tee $OUT/ab_npp_hrd.txt | cut -c1-100
python - <<PY
import json
for wl in ("npp_hrd", "goes_hrit"):
    try:
        for ln in open("$OUT/ab_%s.txt" % wl):
            d = json.loads(ln); k = d["kernels_ms"]
            print(wl, d["cfg"], d["ms_per_step"], d["cadus"], {n: round(v, 2) for n, v in k.items() if n in ("k_afc", "k_mm", "k_quantize", "k_compact8")})
    except Exception as e:
        print(wl, "unreadable", e)
PY
