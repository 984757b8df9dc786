#!/bin/bash
# SQ counters per kernel (VALU issue rate, waits) for one workload
TAG=${1:-rXX}; WL=${2:-goes_hrit}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/sq_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 > $OUT/sq_$WL.log 2>&1
tail -3 $OUT/sq_$WL.log
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/sq2_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 > $OUT/sq2_$WL.log 2>&1
tail -3 $OUT/sq2_$WL.log
python tools/sq_summary.py $OUT $WL | tee $OUT/${WL}_sq.csv
find $OUT -name "*kernel_trace.csv" -size +20M -delete
