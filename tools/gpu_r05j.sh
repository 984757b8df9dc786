#!/bin/bash
# Round 5, visit J (after the closing visit, kernel sources untouched): counters behind DESIGN 7a's claim -- what the lane stages are bound by. SQ counters of the
# driver workload on the final sources (two passes, as round 4's visit Y), the first of them also with the per-lane path (SDHIP_COOP=0), and the vector-memory side:
# TCP / TA counters of k_afc with cooperative and with per-lane access (which names this rocprofv3 knows is listed first)
TAG=${1:-r05_j}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
LEGS="--cpu-samples 0 --others 0 --next-rows 0 --exact-samples 0 --streamed-samples 0 --parity-samples 0"
WL=metop_ahrpt
rocprofv3 -L > $OUT/avail.txt 2>&1; grep -o "TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TCC_[A-Z0-9_]*REQ[A-Z0-9_]*" $OUT/avail.txt | sort -u | tr '\n' ' ' | cut -c1-3000 > $OUT/tcp_names.txt; wc -c $OUT/tcp_names.txt
timeout 280 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/sq_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 $LEGS > $OUT/sq_$WL.log 2>&1
timeout 280 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/sq2_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 $LEGS > $OUT/sq2_$WL.log 2>&1
python tools/sq_summary.py $OUT $WL | head -8 | tee $OUT/${WL}_sq.csv
mkdir -p $OUT/nocoop
SDHIP_COOP=0 timeout 280 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/nocoop/sq_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 $LEGS > $OUT/nocoop_sq.log 2>&1
python tools/sq_summary.py $OUT/nocoop $WL | head -4 | tee $OUT/${WL}_sq_nocoop.csv
for set in "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  for coop in 1 0; do
    SDHIP_COOP=$coop timeout 280 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "k_afc|k_mm" --output-format csv -d $OUT/tcp_${n}_$coop -- python bench.py --workload $WL --steps 1 --warmup 1 $LEGS > $OUT/tcp_${n}_$coop.log 2>&1 || tail -2 $OUT/tcp_${n}_$coop.log
    echo "== $set  SDHIP_COOP=$coop"; python tools/tcp_summary.py $OUT/tcp_${n}_$coop | tee $OUT/tcp_${n}_coop$coop.csv
  done
done
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
