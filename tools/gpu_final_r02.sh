#!/bin/bash
# Final visit of round 2: full parity suite, rocprofv3 kernel stats + the two PMC passes of the MetOp line (stamped with the kernel
# source hash), then the driver's own command lines for all three workloads (CPU legs included).
TAG=${1:-r02_q}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
WL=metop_ahrpt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -- python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0 > $OUT/prof_$WL.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 > $OUT/pmc_${c}_$WL.log 2>&1
done
f=$(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0" > $OUT/metop_kernel_stats.csv && head -16 $OUT/metop_kernel_stats.csv
python tools/pmc_summary.py $OUT $WL > $OUT/metop_pmc.csv 2>&1; head -16 $OUT/metop_pmc.csv
cp $OUT/metop_pmc.csv profiles/${TAG}_metop_pmc.csv
find $OUT -name "*kernel_trace.csv" -size +5M -delete
find $OUT -name "*counter_collection.csv" -size +5M -delete
echo "== driver line"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_metop.json 2> $OUT/bench_metop.err || { echo FAILED; tail -20 $OUT/bench_metop.err; }
head -c 5000 $OUT/bench_metop.json; echo
for WL in goes_hrit npp_hrd; do
  echo "== $WL"; timeout 900 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-samples 20000000 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err || { echo FAILED; tail -20 $OUT/bench_$WL.err; }
  python - <<PY
import json
d=json.load(open("$OUT/bench_$WL.json"))
print({k:d[k] for k in ("value","ms_per_step","soft_parity","cadu_parity")}); print(d["cpu_baseline"]); print(d["roofline"])
PY
done
