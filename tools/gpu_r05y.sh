#!/bin/bash
# Round 5, SECOND closing visit (tag r05_zz) on the final sources after visit M (int8 soft symbols out of the clock recovery by default): the whole GPU suite (SDHIP_FINAL=1: a stale committed PMC profile fails), the driver's own command line, rocprofv3
# kernel stats of the driver workload, the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only) for the three workloads AND the next-row
# benches (DVB-S2 module, DVB-S2 FEC tail, LRPT, FY-3, ndsp chain), kernel stats of the DVB-S2 module / LRPT / FY-3, a short driver-shaped run quoting them
TAG=${1:-r05_zz}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
LEGS="--cpu-samples 0 --others 0 --next-rows 0 --exact-samples 0 --streamed-samples 0"
WL=metop_ahrpt
echo "== PMC passes first (the suite and the bench quote them)"
for WL in metop_ahrpt goes_hrit npp_hrd; do
  S=${WL%%_*}
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 $LEGS > $OUT/pmc_${c}_$WL.log 2>&1
  done
  python tools/pmc_summary.py $OUT $WL > $OUT/${S}_pmc.csv 2>&1; head -6 $OUT/${S}_pmc.csv
  cp $OUT/${S}_pmc.csv profiles/${TAG}_${S}_pmc.csv
done
nr() { tag=$1; shift; for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sdhip" --output-format csv -d $OUT/pmc_${c}_$tag -- "$@" > $OUT/pmc_${c}_$tag.log 2>&1; done
  python tools/pmc_summary.py $OUT $tag > $OUT/${tag}_pmc.csv 2>&1; head -5 $OUT/${tag}_pmc.csv; cp $OUT/${tag}_pmc.csv profiles/${TAG}_${tag}_pmc.csv; }
nr dvbs2 python tools/bench_dvbs2_demod.py --frames 2048 --steps 1 --warmup 1 --cpu-frames 0
nr dvbs2fec python tools/bench_dvbs2.py --steps 1 --cpu-frames 0
nr lrpt python tools/bench_lrpt.py --steps 1 --cpu-frames 0
nr fy3 python tools/bench_fy3.py --steps 1 --cpu-frames 0
nr ndsp python tools/bench_ndsp.py --steps 1 --cpu-samples 1000000
echo "== kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_metop -- python bench.py --workload metop_ahrpt --steps 3 --warmup 1 $LEGS > $OUT/prof_metop.log 2>&1
f=$(find $OUT/prof_metop -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "python bench.py --workload metop_ahrpt --steps 3 --warmup 1 $LEGS" > $OUT/metop_kernel_stats.csv && head -14 $OUT/metop_kernel_stats.csv
ks() { tag=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -- "$@" > $OUT/prof_$tag.log 2>&1; f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "$*" > $OUT/${tag}_kernel_stats.csv && head -8 $OUT/${tag}_kernel_stats.csv; }
find $OUT -name "*kernel_trace.csv" -size +5M -delete
find $OUT -name "*counter_collection.csv" -size +5M -delete
echo "== smoke()"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
echo "== the whole GPU suite"
SDHIP_FINAL=1 timeout 900 python -m pytest tests/ -m gpu -q --durations=12 -k "not full_size" 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt
echo "== the driver's command line"; timeout 1300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench rc $?"; tail -20 $OUT/bench.err; }
python - <<PY
import json
for f in ("$OUT/bench.json",):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, {k:d.get(k) for k in ("value","ms_per_step")}, d.get("roofline"))
    sp = d["soft_parity"]
    print("  soft_parity", sp["frac_within_1e-5"], "arm_grid", sp.get("arm_grid"), "cadu", d["cadu_parity"].get("byte_identical"), "exact", (d.get("exact_mode") or {}).get("value"), "streamed", {k:v for k,v in (d.get("streamed") or {}).items() if k.endswith("GB_per_s")})
    for n,o in (d.get("other_workloads") or {}).items():
        print("  ", n, o.get("ms_per_step"), o.get("roofline"))
    for n,o in (d.get("next_rows") or {}).items():
        print("  next", n, o.get("value"), o.get("unit"), o.get("error"), (o.get("roofline") or {}).get("frac"), (o.get("roofline") or {}).get("traffic"))
PY

echo "== kernel stats of the next rows (unchanged kernels; last, if the budget allows)"
ks dvbs2 python tools/bench_dvbs2_demod.py --frames 2048 --steps 3 --warmup 1 --cpu-frames 0
ks lrpt python tools/bench_lrpt.py --steps 3 --cpu-frames 0
ks fy3 python tools/bench_fy3.py --steps 3 --cpu-frames 0
find $OUT -name "*kernel_trace.csv" -size +5M -delete
