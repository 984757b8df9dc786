#!/bin/bash
# bench the other BASELINE workloads at full size (no CPU leg) + the GPU test-suite
TAG=${1:-rXX}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
for WL in "$@"; do
  SDHIP_DEBUG=1 timeout 600 python bench.py --workload $WL --steps 2 --warmup 1 --cpu-samples 0 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  echo "rc=$?"; tail -c 1500 $OUT/bench_$WL.err; head -c 3000 $OUT/bench_$WL.json; echo
done
