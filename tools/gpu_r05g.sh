#!/bin/bash
# Round 5, visit G: the DVB-S2 module's clock recovery against its warm-up (its loop gain is five times MetOp's time constant: the adaptive warm-up ends at the
# cap, 97 % of the lanes' work) -- SDHIP_W_MM sweeps with the BBFRAME check; the wider PLL branch search at 2e-4 rad/sample; LRPT with the parallel walk
TAG=${1:-r05_g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 200 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-frames 0 > $OUT/dvbs2_$name.json 2> $OUT/dvbs2_$name.err || tail -3 $OUT/dvbs2_$name.err; }
SDHIP_DEBUG=1 timeout 200 python tools/bench_dvbs2_demod.py --frames 2048 --steps 1 --warmup 1 --cpu-frames 0 > $OUT/dvbs2_debug.json 2> $OUT/dvbs2_debug.err; grep -E "mm |k_mm|warm-up" $OUT/dvbs2_debug.err | head -12
run default A=1
run w8k SDHIP_W_MM=8192
run w12k SDHIP_W_MM=12288
run w22k SDHIP_W_MM=22016
run w12k_l16k SDHIP_W_MM=12288 SDHIP_LANES_MM=16384
timeout 200 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-frames 0 --cfo-rad 2e-4 > $OUT/dvbs2_cfo2.json 2> $OUT/dvbs2_cfo2.err || tail -3 $OUT/dvbs2_cfo2.err
timeout 200 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-frames 0 --cfo-rad 4e-4 > $OUT/dvbs2_cfo4.json 2> $OUT/dvbs2_cfo4.err || tail -3 $OUT/dvbs2_cfo4.err
timeout 200 python tools/bench_lrpt.py > $OUT/bench_lrpt.json 2> $OUT/bench_lrpt.err || tail -5 $OUT/bench_lrpt.err
timeout 300 python -m pytest tests/test_lrpt_gpu.py tests/test_dvbs2_gpu.py -m gpu -q -x -k "lrpt or engine" 2>&1 | tail -4 | tee $OUT/pytest_sel.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    k = d.get("kernels_ms") or {}
    print(f.split("/")[-1], d.get("value"), d.get("unit"), "ms", d.get("ms_per_step"), "in_order", d.get("all_bbframes_are_transmitted_ones_in_order"), "bad", d.get("frames_not_matching"), "frames", d.get("bbframes_per_step"),
          "k_mm", k.get("k_mm"), "rerun", k.get("k_mm (re-run launches, included in k_mm)"), "pll", (d.get("pll_schedule_per_step") or {}).get("forced"), dict(list(k.items())[:3]) if "lrpt" in f else "")
PY
