#!/bin/bash
# Round 5, visit M: the clock recovery storing int8 soft symbols as the default (eight symbols per 16-byte store out of a shift register, branch-free clamp) and the
# wide compaction behind it (k_compact8: two aligned 16-byte loads through a funnel shift, one aligned store) -- the byte-identity tests on the device, then A/B
# against the float rows + k_quantize (SDHIP_MM_Q8=0) on the three workloads
TAG=${1:-r05_m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zy_demod_additions_gpu.py tests/test_golden_gpu.py -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_q8.txt
timeout 300 python tools/ab_demod.py --workload metop_ahrpt --steps 4 --warmup 2 "" "SDHIP_MM_Q8=0" "" "SDHIP_MM_Q8=0" 2> $OUT/ab_metop_ahrpt.err | tee $OUT/ab_metop_ahrpt.txt | cut -c1-200
timeout 200 python tools/ab_demod.py --workload goes_hrit --steps 4 --warmup 2 "" "SDHIP_MM_Q8=0" 2> $OUT/ab_goes_hrit.err | tee $OUT/ab_goes_hrit.txt | cut -c1-200
python - <<PY
import json
for wl in ("metop_ahrpt", "npp_hrd", "goes_hrit"):
    try:
        for ln in open("$OUT/ab_%s.txt" % wl):
            d = json.loads(ln); k = d["kernels_ms"]
            print(wl, d["cfg"], d["ms_per_step"], d["cadus"], {n: round(v, 2) for n, v in k.items() if n in ("k_afc", "k_mm", "k_quantize", "k_compact8")})
    except Exception as e:
        print(wl, "unreadable", e)
PY
