#!/bin/bash
# Round 5, visit F: the whole GPU suite but the two full-size reference decodes (left to the closing visit), then the next-row benches that changed: the DVB-S2
# module with a carrier offset and freq_prop_factor 0.01 (does the header-anchored PLL hold it, what the reference chain does beside it), LRPT (parallel chain), FY-3
TAG=${1:-r05_f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -m gpu -q -k "not full_size_metop and not full_size_npp" --durations=8 2>&1 | tail -22 | tee $OUT/pytest_gpu.txt
timeout 400 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-procs 0 > $OUT/bench_dvbs2_demod.json 2> $OUT/bench_dvbs2_demod.err || tail -5 $OUT/bench_dvbs2_demod.err
timeout 200 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-frames 0 --cfo-rad 2e-4 > $OUT/bench_dvbs2_demod_cfo2.json 2> $OUT/bench_dvbs2_demod_cfo2.err || tail -5 $OUT/bench_dvbs2_demod_cfo2.err
timeout 200 python tools/bench_lrpt.py > $OUT/bench_lrpt.json 2> $OUT/bench_lrpt.err || tail -5 $OUT/bench_lrpt.err
timeout 200 python tools/bench_fy3.py > $OUT/bench_fy3.json 2> $OUT/bench_fy3.err || tail -5 $OUT/bench_fy3.err
python - <<PY
import json
for f in ("bench_dvbs2_demod", "bench_dvbs2_demod_cfo2", "bench_lrpt", "bench_fy3"):
    try:
        d = json.loads(open("$OUT/" + f + ".json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d.get("value"), d.get("unit"), "ms", d.get("ms_per_step"), {k: d.get(k) for k in ("all_bbframes_are_transmitted_ones_in_order", "frames_not_matching", "bbframes_per_step") if k in d})
    print("   kernels", dict(list((d.get("kernels_ms") or {}).items())[:8]))
    if "parity_sample" in d:
        p = d["parity_sample"]; print("   parity", {k: p[k] for k in p if k not in ("acquisition",)})
    if "pll_schedule_per_step" in d: print("   pll", d["pll_schedule_per_step"], d.get("stats", {}).get("freq_hz"))
PY
