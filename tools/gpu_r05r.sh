#!/bin/bash
# Round 5, visit R (final sources): visit J's vector-memory counters for the clock recovery with its int8 rows (k_mm<.., Q8>) and k_compact8
TAG=${1:-r05_r}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
LEGS="--cpu-samples 0 --others 0 --next-rows 0 --exact-samples 0 --streamed-samples 0 --parity-samples 0"
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 100 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "k_mm|k_compact8|k_afc" --output-format csv -d $OUT/tcp_${n} -- python bench.py --workload metop_ahrpt --steps 1 --warmup 1 $LEGS > $OUT/tcp_${n}.log 2>&1 || tail -2 $OUT/tcp_${n}.log
  echo "== $set"; python tools/tcp_summary.py $OUT/tcp_${n} | tee $OUT/tcp_${n}.csv
done
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
