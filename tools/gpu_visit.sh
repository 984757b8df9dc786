#!/bin/bash
# ONE script for every GPU-box visit of a round (replaces the 70 per-visit scripts of rounds 2-5, which live on in git history):
#   tools/gpu_visit.sh <tag> <stage> [<stage> ...]
# Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/. Stages:
#   suite            pytest -m gpu (whole suite), tail -> pytest_gpu.txt          (SDHIP_FINAL=1 in the environment: the closing run)
#   tests:<expr>     pytest -m gpu -k <expr>
#   smoke            __graft_entry__.smoke()
#   driver           python bench.py --gpus 1 --steps 20 --warmup 5 (the driver's own command) -> bench_line.json + bench_detail.json
#   quick:<wl>       bench.py on one workload, parity over 400 M samples, no other rows (~1.5 min) -> quick_<wl>.json (the full object); extra SDHIP_* env applies
#   ab:<wl>:<VAR=v,VAR=v>   the same under the given environment -> ab_<wl>_<VAR=v,..>.json
#   stats:<wl>       rocprofv3 --kernel-trace --stats of 3 timed steps -> <wl>_kernel_stats.csv
#   pmc:<wl>         the two PMC passes (FETCH_SIZE, WRITE_SIZE apart, per the microarch guide) -> <wl>_pmc.csv (stamped with the kernel sources' hash)
#   sq:<wl>          the two SQ counter passes (VALU issue, waits, LDS) of one workload -> <wl>_sq.csv (tools/sq_summary.py)
#   row:<tool>[:args]  one of tools/bench_{ndsp,dvbs2,dvbs2_demod,lrpt,fy3}.py -> row_<tool>.json
#   rowpmc:<tag>:<tool>[:args]    the two PMC passes of a row tool -> <tag>_pmc.csv (tag = what the tool's roofline looks up: dvbs2, dvbs2fec, lrpt, fy3, ndsp)
#   rowstats:<tag>:<tool>[:args]  rocprofv3 --kernel-trace --stats of a row tool -> <tag>_kernel_stats.csv
#   publish          copies the visit's *_pmc.csv to profiles/<tag>_<short>_pmc.csv on the box (the closing visit: PMC first, publish, then suite + driver)
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
QUICK="--others 0 --next-rows 0 --exact-samples 0 --streamed-samples 0 --parity-samples 400000000 --steps 6 --warmup 2"
for ST in "$@"; do
  echo "=== $ST"
  case $ST in
    suite)   python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt ;;
    tests:*) python -m pytest tests -m gpu -x -q -k "${ST#tests:}" 2>&1 | tail -8 | tee -a $OUT/pytest_some.txt ;;
    smoke)   python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt ;;
    driver)  python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_detail.json > $OUT/bench_line.json 2> $OUT/bench.err; echo rc=$?; wc -c $OUT/bench_line.json; cat $OUT/bench_line.json; tail -3 $OUT/bench.err ;;
    quick:*) WL=${ST#quick:}; python bench.py --workload $WL $QUICK --detail $OUT/quick_$WL.json > $OUT/quick_$WL.line 2> $OUT/quick_$WL.err; echo rc=$?; cat $OUT/quick_$WL.line; tail -3 $OUT/quick_$WL.err ;;
    ab:*)    R=${ST#ab:}; WL=${R%%:*}; ENVS=${R#*:}; ( for kv in ${ENVS//,/ }; do export "$kv"; done; python bench.py --workload $WL $QUICK --detail $OUT/ab_${WL}_$ENVS.json > $OUT/ab_${WL}_$ENVS.line 2> $OUT/ab_${WL}_$ENVS.err; echo rc=$?; cat $OUT/ab_${WL}_$ENVS.line; tail -3 $OUT/ab_${WL}_$ENVS.err ) ;;
    stats:*) WL=${ST#stats:}; CMD="python bench.py --workload $WL --steps 3 --warmup 1 --cpu-samples 0 --others 0 --next-rows 0"
             rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -- $CMD > $OUT/prof_$WL.log 2>&1
             f=$(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "$CMD" > $OUT/${WL}_kernel_stats.csv && head -14 $OUT/${WL}_kernel_stats.csv
             rm -rf $OUT/prof_$WL ;;
    pmc:*)   WL=${ST#pmc:}
             for c in FETCH_SIZE WRITE_SIZE; do
               timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 --others 0 --next-rows 0 --exact-samples 0 --streamed-samples 0 > $OUT/pmc_${c}_$WL.log 2>&1
             done
             python tools/pmc_summary.py $OUT $WL > $OUT/${WL}_pmc.csv 2>&1; head -30 $OUT/${WL}_pmc.csv
             rm -rf $OUT/pmc_FETCH_SIZE_$WL $OUT/pmc_WRITE_SIZE_$WL ;;  # the raw traces are tens of MB: gpurun copies back at most 64 MiB
    sq:*)    WL=${ST#sq:}; B="python bench.py --workload $WL --steps 1 --warmup 1 --cpu-samples 0 --others 0 --next-rows 0"
             rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/sq_$WL -- $B > $OUT/sq_$WL.log 2>&1
             rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/sq2_$WL -- $B > $OUT/sq2_$WL.log 2>&1
             python tools/sq_summary.py $OUT $WL | tee $OUT/${WL}_sq.csv; rm -rf $OUT/sq_$WL $OUT/sq2_$WL ;;
    row:*)   R=${ST#row:}; T=${R%%:*}; A=""; [ "$R" != "$T" ] && A=${R#*:}; python tools/bench_$T.py ${A//,/ } > $OUT/row_$T.json 2> $OUT/row_$T.err; echo rc=$?; head -c 3000 $OUT/row_$T.json; echo; tail -3 $OUT/row_$T.err ;;
    rowpmc:*) R=${ST#rowpmc:}; TG=${R%%:*}; R2=${R#*:}; T=${R2%%:*}; A=""; [ "$R2" != "$T" ] && A=${R2#*:}
             for c in FETCH_SIZE WRITE_SIZE; do
               timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$TG -- python tools/bench_$T.py ${A//,/ } > $OUT/pmc_${c}_$TG.log 2>&1
             done
             python tools/pmc_summary.py $OUT $TG > $OUT/${TG}_pmc.csv 2>&1; head -6 $OUT/${TG}_pmc.csv
             rm -rf $OUT/pmc_FETCH_SIZE_$TG $OUT/pmc_WRITE_SIZE_$TG ;;
    rowstats:*) R=${ST#rowstats:}; TG=${R%%:*}; R2=${R#*:}; T=${R2%%:*}; A=""; [ "$R2" != "$T" ] && A=${R2#*:}; CMD="python tools/bench_$T.py ${A//,/ }"
             rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TG -- $CMD > $OUT/prof_$TG.log 2>&1
             f=$(find $OUT/prof_$TG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "$CMD" > $OUT/${TG}_kernel_stats.csv && head -8 $OUT/${TG}_kernel_stats.csv
             rm -rf $OUT/prof_$TG ;;
    publish) # the PMC summaries taken so far -> profiles/<tag>_<short>_pmc.csv ON THE BOX, so that the suite / bench stages behind it quote them (locally: copy the same files)
             for f in $OUT/*_pmc.csv; do b=$(basename $f _pmc.csv); b=${b/metop_ahrpt/metop}; b=${b/goes_hrit/goes}; b=${b/npp_hrd/npp}; cp $f profiles/${TAG}_${b}_pmc.csv; echo "profiles/${TAG}_${b}_pmc.csv"; done ;;
    *) echo "unknown stage $ST" ;;
  esac
done
