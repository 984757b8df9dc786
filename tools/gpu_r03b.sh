#!/bin/bash
# Round 3, visit B: DVB-S2 LDPC parity tests on the GPU, k_mm changes (A/B against the previous switches is moot: sources changed),
# kernel stats + SQ counters of the new demodulator kernels, the MetOp line without the long reference leg.
TAG=${1:-r03_b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dvbs2_gpu.py -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest_dvbs2.txt; tail -5 $OUT/pytest_dvbs2.txt
timeout 900 python -m pytest tests/test_demod_gpu.py tests/test_golden_gpu.py tests/test_zy_demod_additions_gpu.py -m gpu -q -x -k "not full_size" 2>&1 | tail -6 > $OUT/pytest_demod.txt; tail -3 $OUT/pytest_demod.txt
timeout 600 python tools/ab_demod.py --workload metop_ahrpt "" "SDHIP_LANES_MM=130560" "SDHIP_LANES_MM=98304" "SDHIP_COSTAS_TAUS=16" > $OUT/ab_metop.txt 2> $OUT/ab_metop.err; cat $OUT/ab_metop.txt; tail -3 $OUT/ab_metop.err
timeout 300 python tools/ab_demod.py --workload npp_hrd "" > $OUT/ab_npp.txt 2> $OUT/ab_npp.err; cat $OUT/ab_npp.txt
WL=metop_ahrpt
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --kernel-include-regex "k_afc|k_mm|k_ldpc" --output-format csv -d $OUT/sq_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 --others 0 > $OUT/sq_$WL.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "k_afc|k_mm|k_ldpc" --output-format csv -d $OUT/sq2_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 --others 0 > $OUT/sq2_$WL.log 2>&1
python tools/sq_summary.py $OUT $WL | tee $OUT/${WL}_sq.csv | head -8
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
