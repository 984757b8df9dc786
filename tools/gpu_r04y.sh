#!/bin/bash
# Round 4, visit Y (after the closing visit, same sources): SQ counters per kernel of the driver workload -- VALU issue, waits, LDS, SALU, VMEM --
# in two --pmc passes (kernel trace only), what bounds k_afc / k_mm / k_vit2h_acs on the final sources
TAG=${1:-r04_y}; WL=${2:-metop_ahrpt}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
LEGS="--cpu-samples 0 --others 0 --next-rows 0 --exact-samples 0 --streamed-samples 0 --parity-samples 0"
timeout 280 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/sq_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 $LEGS > $OUT/sq_$WL.log 2>&1
tail -2 $OUT/sq_$WL.log | cut -c1-300
timeout 280 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/sq2_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 $LEGS > $OUT/sq2_$WL.log 2>&1
tail -2 $OUT/sq2_$WL.log | cut -c1-300
python tools/sq_summary.py $OUT $WL | tee $OUT/${WL}_sq.csv
find $OUT -name "*kernel_trace.csv" -size +5M -delete
find $OUT -name "*counter_collection.csv" -size +5M -delete
