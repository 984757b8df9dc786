#!/bin/bash
# After the batched punctured run: the FEC parity tests (all decoders) on the GPU, then -- only if they are green -- the two PMC
# passes of the MetOp line on these sources
TAG=${1:-r02_s}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 110 python -m pytest tests/test_zz_punctured_gpu.py tests/test_fec_gpu.py -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_fec.txt; tail -3 $OUT/pytest_fec.txt
grep -q " passed" $OUT/pytest_fec.txt && ! grep -q "failed\|error" $OUT/pytest_fec.txt || { echo "NOT GREEN"; exit 1; }
WL=metop_ahrpt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 > $OUT/pmc_${c}_$WL.log 2>&1
done
python tools/pmc_summary.py $OUT $WL > $OUT/metop_pmc.csv 2>&1; head -6 $OUT/metop_pmc.csv
find $OUT -name "*kernel_trace.csv" -size +5M -delete
find $OUT -name "*counter_collection.csv" -size +5M -delete
