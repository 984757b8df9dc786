#!/bin/bash
# Round 4, closing visit on the final sources: the whole GPU suite, the driver's own command line, rocprofv3 kernel stats of the driver workload, the two PMC
# passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only) for all three workloads stamped with the source hash, a short driver-shaped run quoting them
TAG=${1:-r04_z}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -m gpu -q --durations=15 2>&1 | tail -45 | tee $OUT/pytest_gpu.txt
echo "== the driver's command line"; timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench rc $?"; tail -20 $OUT/bench.err; }
LEGS="--cpu-samples 0 --others 0 --next-rows 0 --exact-samples 0 --streamed-samples 0"
WL=metop_ahrpt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -- python bench.py --workload $WL --steps 3 --warmup 1 $LEGS > $OUT/prof_$WL.log 2>&1
f=$(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "python bench.py --workload $WL --steps 3 --warmup 1 $LEGS" > $OUT/metop_kernel_stats.csv && head -14 $OUT/metop_kernel_stats.csv
for WL in metop_ahrpt goes_hrit npp_hrd; do
  S=${WL%%_*}
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 $LEGS > $OUT/pmc_${c}_$WL.log 2>&1
  done
  python tools/pmc_summary.py $OUT $WL > $OUT/${S}_pmc.csv 2>&1; head -8 $OUT/${S}_pmc.csv
  cp $OUT/${S}_pmc.csv profiles/${TAG}_${S}_pmc.csv
done
find $OUT -name "*kernel_trace.csv" -size +5M -delete
find $OUT -name "*counter_collection.csv" -size +5M -delete
echo "== short driver-shaped run (quotes the PMC traffic of the profiles just taken)"
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --parity-samples 200000000 --next-rows 0 --exact-samples 0 --streamed-samples 0 > $OUT/bench_short.json 2> $OUT/bench_short.err || { echo "bench rc $?"; tail -20 $OUT/bench_short.err; }
python - <<PY
import json
for f in ("$OUT/bench.json", "$OUT/bench_short.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, {k:d.get(k) for k in ("value","ms_per_step")}, d.get("roofline"))
    print("  soft_parity", d["soft_parity"]["frac_within_1e-5"], "cadu", d["cadu_parity"].get("byte_identical"), "exact", (d.get("exact_mode") or {}).get("value"), "streamed", {k:v for k,v in (d.get("streamed") or {}).items() if k.endswith("GB_per_s")})
    for n,o in (d.get("other_workloads") or {}).items():
        print("  ", n, o.get("ms_per_step"), o.get("roofline"))
    for n,o in (d.get("next_rows") or {}).items():
        print("  next", n, o.get("value"), o.get("unit"), o.get("error"))
PY
