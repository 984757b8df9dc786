#!/bin/bash
# shortest useful GPU visit: rocprofv3 kernel stats of the bench command, the bench line itself, then the auto-geometry tests
TAG=${1:-rXX}; WL=goes_hrit
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -- python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0 > $OUT/prof_$WL.log 2>&1
f=$(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0" > $OUT/${WL}_kernel_stats.csv && head -9 $OUT/${WL}_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
python bench.py --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err; head -c 900 $OUT/bench_$WL.json; echo
python -m pytest tests/test_demod_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest_demod_gpu.txt
