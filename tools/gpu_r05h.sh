#!/bin/bash
# Round 5, visit H: the DVB-S2 module at larger carrier offsets with the header-rate branch pick and the guarded hand-over; the LDPC rows (worst case / converging)
TAG=${1:-r05_h}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for c in 1e-4 2e-4 4e-4; do
  SDHIP_DEBUG=1 timeout 200 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --cpu-frames 0 --cfo-rad $c > $OUT/dvbs2_cfo_$c.json 2> $OUT/dvbs2_cfo_$c.err || tail -3 $OUT/dvbs2_cfo_$c.err
  grep -E "s2 pll: (rate|[0-9]+ of)" $OUT/dvbs2_cfo_$c.err | head -6
done
timeout 200 python tools/bench_dvbs2.py --rate 2/3 --sigma 13 --cpu-frames 0 > $OUT/dvbs2_fec.json 2> $OUT/dvbs2_fec.err || tail -3 $OUT/dvbs2_fec.err
timeout 200 python tools/bench_dvbs2.py --rate 2/3 --front 0 --sigma 10.5 --sync-frames 0 --cpu-frames 0 > $OUT/dvbs2_fec_conv.json 2> $OUT/dvbs2_fec_conv.err || tail -3 $OUT/dvbs2_fec_conv.err
timeout 200 python tools/bench_dvbs2.py --rate 2/3 --front 0 --sigma 12 --sync-frames 0 --cpu-frames 0 > $OUT/dvbs2_fec_conv12.json 2> $OUT/dvbs2_fec_conv12.err || tail -3 $OUT/dvbs2_fec_conv12.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    k = d.get("kernels_ms") or {}
    print(f.split("/")[-1], d.get("value"), d.get("unit"), "ms", d.get("ms_per_step"), "in_order", d.get("all_bbframes_are_transmitted_ones_in_order"), "bad", d.get("frames_not_matching"),
          "frames", d.get("bbframes_per_step"), "forced", (d.get("pll_schedule_per_step") or {}).get("forced"), "freq_hz", (d.get("stats") or {}).get("freq_hz"),
          "conv", d.get("frames_converged"), "upd", d.get("update_passes_per_frame"), "ldpc", k.get("k_ldpc_trial"), (d.get("roofline") or {}).get("frac"))
PY
