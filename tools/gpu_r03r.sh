#!/bin/bash
# Round 3, last visit: the DVB-S2 synchroniser / PLL / front-end tests on the GPU, kernel stats + the two PMC passes of the FINAL sources on the driver
# workload alone (--others 0 --next-rows 0), a short driver-shaped run that quotes that PMC traffic, the DVB-S2 chain with its demapper stage
TAG=${1:-r03_r}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_dvbs2_gpu.py tests/test_zy_demod_additions_gpu.py -m gpu -q -k "pl_sync or pll or atan2f or bbframes or bb_to_soft or dvbs2_front" 2>&1 | tail -6 | tee $OUT/pytest_new.txt
WL=metop_ahrpt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -- python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0 --others 0 --next-rows 0 > $OUT/prof_$WL.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 --others 0 --next-rows 0 > $OUT/pmc_${c}_$WL.log 2>&1
done
f=$(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0 --others 0 --next-rows 0" > $OUT/metop_kernel_stats.csv && head -14 $OUT/metop_kernel_stats.csv
python tools/pmc_summary.py $OUT $WL > $OUT/metop_pmc.csv 2>&1; head -12 $OUT/metop_pmc.csv
cp $OUT/metop_pmc.csv profiles/${TAG}_metop_pmc.csv
find $OUT -name "*kernel_trace.csv" -size +5M -delete
find $OUT -name "*counter_collection.csv" -size +5M -delete
echo "== short driver-shaped run"; timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --parity-samples 200000000 --others 0 --next-rows 0 > $OUT/bench_short.json 2> $OUT/bench.err || { echo "bench rc $?"; tail -20 $OUT/bench.err; }
python - <<PY
import json
d=json.load(open("$OUT/bench_short.json"))
print({k:d[k] for k in ("value","ms_per_step","roofline")})
print("soft_parity", d["soft_parity"]["frac_within_1e-5"], "cadu", d["cadu_parity"]["byte_identical"], d["cadu_parity"]["compared"])
print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}" + (f"(x{v['traffic_over_algorithmic']})" if 'traffic_over_algorithmic' in v else '') for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:10]))
PY
