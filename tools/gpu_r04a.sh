#!/bin/bash
# Round 4, visit A: the DVB-S2 demodulator handle and its frame-parallel PLL on the GPU (new tests), the plugin's dvbs2_demod module through the
# minihost, BASELINE configs[4] measured (tools/bench_dvbs2_demod.py) with per-kernel HIP-event times + a rocprofv3 kernel-stats pass of the same command
TAG=${1:-r04_a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dvbs2_gpu.py tests/test_plugin_minihost_gpu.py -m gpu -q -k "pll_parallel or engine or mirror or dvbs2_module or test_pll or bbframes" 2>&1 | tail -15 | tee $OUT/pytest_new.txt
echo "== bench_dvbs2_demod"
timeout 900 python tools/bench_dvbs2_demod.py --frames 512 --steps 3 > $OUT/bench_dvbs2_demod.json 2> $OUT/bench_dvbs2_demod.err || { echo "rc $?"; tail -30 $OUT/bench_dvbs2_demod.err; }
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_dvbs2_demod.json"))
    for k in ("value","unit","Msamples_per_s","frames_per_s","realtime_factor_at_45_Msym_per_s","ms_per_step","bbframes_per_step","all_bbframes_are_transmitted_ones_in_order","frames_not_matching","pll_schedule_per_step","kernels_ms","roofline","whole_path","cpu_baseline","parity_sample","stats"):
        print(k, d.get(k))
except Exception as e:
    print("no result", e)
PY
echo "== exact mode (serial schedules), small"
timeout 600 python tools/bench_dvbs2_demod.py --frames 64 --steps 1 --exact 1 --cpu-frames 0 > $OUT/bench_dvbs2_demod_exact.json 2> $OUT/bench_dvbs2_demod_exact.err || { echo "rc $?"; tail -30 $OUT/bench_dvbs2_demod_exact.err; }
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_dvbs2_demod_exact.json"))
    for k in ("value","ms_per_step","bbframes_per_step","all_bbframes_are_transmitted_ones_in_order","kernels_ms"):
        print(k, d.get(k))
except Exception as e:
    print("no result", e)
PY
echo "== rocprofv3 kernel stats of the bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python tools/bench_dvbs2_demod.py --frames 512 --steps 3 --cpu-frames 0 > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f "python tools/bench_dvbs2_demod.py --frames 512 --steps 3 --cpu-frames 0" > $OUT/dvbs2_demod_kernel_stats.csv && head -24 $OUT/dvbs2_demod_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +5M -delete
