#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprofv3 kernel stats, and the two PMC passes (FETCH_SIZE / WRITE_SIZE apart,
# per the microarch guide). Usage: tools/gpu_round.sh <tag> [workload]
TAG=${1:-rXX}; WL=${2:-goes_hrit}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
python bench.py --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err; tail -c 600 $OUT/bench_$WL.err; head -c 1500 $OUT/bench_$WL.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -- python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0 > $OUT/prof_$WL.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 > $OUT/pmc_${c}_$WL.log 2>&1
done
find $OUT -name "*.csv" | head -20
f=$(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0" > $OUT/${WL}_kernel_stats.csv && head -12 $OUT/${WL}_kernel_stats.csv
python tools/pmc_summary.py $OUT $WL > $OUT/${WL}_pmc.csv 2>&1; head -30 $OUT/${WL}_pmc.csv
[ -z "$NO_SQ" ] && { tools/gpu_sq.sh $TAG $WL > /dev/null 2>&1; head -12 $OUT/${WL}_sq.csv; }
# keep only the small summaries in the merge-back
find $OUT -name "*kernel_trace.csv" -size +20M -delete
