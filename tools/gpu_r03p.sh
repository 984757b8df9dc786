#!/bin/bash
# Round 3, full visit: the whole GPU suite, kernel stats + the two PMC passes of the current sources on the driver workload, the driver line (with other_workloads and next_rows)
TAG=${1:-r03_p}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -22 > $OUT/pytest_gpu.txt; tail -12 $OUT/pytest_gpu.txt
WL=metop_ahrpt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -- python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0 --others 0 > $OUT/prof_$WL.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 --others 0 > $OUT/pmc_${c}_$WL.log 2>&1
done
f=$(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0 --others 0" > $OUT/metop_kernel_stats.csv && head -14 $OUT/metop_kernel_stats.csv
python tools/pmc_summary.py $OUT $WL > $OUT/metop_pmc.csv 2>&1; head -12 $OUT/metop_pmc.csv
cp $OUT/metop_pmc.csv profiles/${TAG}_metop_pmc.csv
find $OUT -name "*kernel_trace.csv" -size +5M -delete
find $OUT -name "*counter_collection.csv" -size +5M -delete
echo "== driver line"; timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench rc $?"; tail -20 $OUT/bench.err; }
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","roofline")})
print("soft_parity", {k:v for k,v in d["soft_parity"].items() if k!="what"}); print("cadu_parity", {k:v for k,v in d["cadu_parity"].items() if k!="tail"}); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"])
print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]))
for k,v in d.get("other_workloads",{}).items(): print(k, {a:v[a] for a in ("value","ms_per_step","top_kernels_ms")}, v["soft_parity"]["frac_within_1e-5"], v["cadu_parity"]["byte_identical"], v["cadu_parity"]["compared"])
PY
python -c "import json; d=json.load(open(\"$OUT/bench.json\")); print(json.dumps(d.get(\"next_rows\"))[:1500])"
