#!/bin/bash
# Round 5, visit I (before the closing visit): the DVB-S2 module with its own clock-recovery windows (time, BBFRAMEs at 1e-4 and 4e-4 rad/sample, psk_demod's windows
# beside them), the whole GPU suite but the two full-size reference decodes, a short driver-shaped run (validates the bench line's new members)
TAG=${1:-r05_i}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
SDHIP_DEBUG=1 timeout 200 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-frames 0 > $OUT/dvbs2_default.json 2> $OUT/dvbs2_default.err || tail -3 $OUT/dvbs2_default.err
grep -E "\] mm " $OUT/dvbs2_default.err | head -6
timeout 200 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-frames 0 --cfo-rad 4e-4 > $OUT/dvbs2_cfo4.json 2> $OUT/dvbs2_cfo4.err || tail -3 $OUT/dvbs2_cfo4.err
SDHIP_S2_MM_TIGHT_MILLI=0 timeout 200 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-frames 0 > $OUT/dvbs2_oldwindows.json 2> $OUT/dvbs2_oldwindows.err || tail -3 $OUT/dvbs2_oldwindows.err
SDHIP_S2_MM_TIGHT_MILLI=8 SDHIP_S2_MM_TOL_MILLI=20 timeout 200 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-frames 0 > $OUT/dvbs2_w8_20.json 2> $OUT/dvbs2_w8_20.err || tail -3 $OUT/dvbs2_w8_20.err
timeout 400 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-procs 0 > $OUT/dvbs2_parity.json 2> $OUT/dvbs2_parity.err || tail -3 $OUT/dvbs2_parity.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/dvbs2_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    k = d.get("kernels_ms") or {}
    print(f.split("/")[-1], d.get("value"), d.get("unit"), "ms", d.get("ms_per_step"), "in_order", d.get("all_bbframes_are_transmitted_ones_in_order"), "bad", d.get("frames_not_matching"),
          "frames", d.get("bbframes_per_step"), "forced", (d.get("pll_schedule_per_step") or {}).get("forced"), "k_mm", k.get("k_mm"), "rerun", k.get("k_mm (re-run launches, included in k_mm)"))
    if "parity_sample" in d:
        p = d["parity_sample"]; print("   parity", {kk: p[kk] for kk in ("reference_frames", "reference_frames_that_are_transmitted_ones", "our_frames_that_are_transmitted_ones_on_the_same_positions", "first_transmitted_frame", "frames_compared", "byte_identical")})
PY
timeout 900 python -m pytest tests/ -m gpu -q -k "not full_size_metop and not full_size_npp" 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 3 --warmup 1 --parity-samples 100000000 --others 0 --exact-samples 20000000 --streamed-samples 0 > $OUT/bench_short.json 2> $OUT/bench_short.err || { echo "bench rc $?"; tail -12 $OUT/bench_short.err; }
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_short.json").read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step")}, d.get("roofline"))
    sp=d["soft_parity"]; print("soft_parity", sp["frac_within_1e-5"], "arm_grid", sp.get("arm_grid"))
    print("cadu", d["cadu_parity"].get("byte_identical"), "gates", d.get("gates"))
    for n,o in (d.get("next_rows") or {}).items():
        print("  next", n, o.get("value"), o.get("unit"), o.get("error"), (o.get("roofline") or {}).get("kernel"), (o.get("roofline") or {}).get("frac"))
except Exception as e:
    print("bench_short unreadable", e)
PY
