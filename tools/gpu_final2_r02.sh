#!/bin/bash
# Last visit of round 2 (after the quantiser / MetOp prep changes): parity suite, kernel stats + PMC passes of the final sources, the driver line
TAG=${1:-r02_r}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
WL=metop_ahrpt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -- python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0 > $OUT/prof_$WL.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 > $OUT/pmc_${c}_$WL.log 2>&1
done
f=$(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0" > $OUT/metop_kernel_stats.csv && head -12 $OUT/metop_kernel_stats.csv
python tools/pmc_summary.py $OUT $WL > $OUT/metop_pmc.csv 2>&1; head -10 $OUT/metop_pmc.csv
cp $OUT/metop_pmc.csv profiles/${TAG}_metop_pmc.csv
find $OUT -name "*kernel_trace.csv" -size +5M -delete
find $OUT -name "*counter_collection.csv" -size +5M -delete
echo "== driver line"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_metop.json 2> $OUT/bench_metop.err || { echo FAILED; tail -20 $OUT/bench_metop.err; }
python - <<PY
import json
d=json.load(open("$OUT/bench_metop.json"))
print({k:d[k] for k in ("value","ms_per_step","soft_parity","cadu_parity","roofline")})
print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]))
PY
for WL in goes_hrit npp_hrd; do
  timeout 300 python bench.py --workload $WL --steps 6 --warmup 2 --cpu-samples 0 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err || { echo FAILED $WL; tail -5 $OUT/bench_$WL.err; }
  python - <<PY
import json
d=json.load(open("$OUT/bench_$WL.json"))
print("$WL", d["value"], d["ms_per_step"], d["check"]["cadus_matching_transmitted"])
print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:10]))
PY
done
