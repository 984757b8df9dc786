#!/bin/bash
# Round-2 GPU visit: parity suite, small bench lines of every workload (JSON plumbing incl. the parity + CPU legs), the
# full-size default line (MetOp, BASELINE configs[2]) and its rocprofv3 kernel stats + PMC passes, GOES full size beside it.
# Usage: tools/gpu_r02.sh <tag> [quick]
TAG=${1:-r02_b}; QUICK=$2
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
for WL in goes_hrit npp_hrd metop_ahrpt; do
  F=$([ $WL = goes_hrit ] && echo 618 || echo 1008)
  echo "== small $WL"; SDHIP_DEBUG=1 timeout 600 python bench.py --workload $WL --frames $F --cpu-samples 4000000 > $OUT/small_$WL.json 2> $OUT/small_$WL.err || { echo FAILED; tail -20 $OUT/small_$WL.err; }
  python - <<PY
import json
try:
    d=json.load(open("$OUT/small_$WL.json"))
    print({k:d[k] for k in ("value","ms_per_step","soft_parity","cadu_parity","check")}); print(d["cpu_baseline"])
except Exception as e: print("no json", e)
PY
done
[ -n "$QUICK" ] && exit 0
for WL in metop_ahrpt goes_hrit; do
  echo "== full $WL"; SDHIP_DEBUG=1 timeout 900 python bench.py --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err || { echo FAILED; tail -30 $OUT/bench_$WL.err; }
  grep -v "host wall" $OUT/bench_$WL.err | tail -25; head -c 6000 $OUT/bench_$WL.json; echo
done
WL=metop_ahrpt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -- python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0 > $OUT/prof_$WL.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 > $OUT/pmc_${c}_$WL.log 2>&1
done
f=$(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0" > $OUT/${WL}_kernel_stats.csv && head -14 $OUT/${WL}_kernel_stats.csv
python tools/pmc_summary.py $OUT $WL > $OUT/${WL}_pmc.csv 2>&1; head -16 $OUT/${WL}_pmc.csv
find $OUT -name "*kernel_trace.csv" -size +20M -delete
