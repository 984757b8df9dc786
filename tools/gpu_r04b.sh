#!/bin/bash
# Round 4, visit B: PLL branch fix re-checked (the three tests visit A failed), BASELINE configs[4] at 2048 frames per step, the host path (streamed) and
# exact-mode legs of bench.py on the driver workload, host_path_rate
TAG=${1:-r04_b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dvbs2_gpu.py tests/test_plugin_minihost_gpu.py -m gpu -q -k "pll_parallel or engine or dvbs2_module" 2>&1 | tail -8 | tee $OUT/pytest_new.txt
timeout 600 python -m pytest tests/test_demod_gpu.py tests/test_zy_demod_additions_gpu.py -m gpu -q -x -k "not full_size and not margin" 2>&1 | tail -4 | tee $OUT/pytest_demod.txt
echo "== bench_dvbs2_demod (2048 frames)"
timeout 900 python tools/bench_dvbs2_demod.py --steps 3 > $OUT/bench_dvbs2_demod.json 2> $OUT/bench_dvbs2_demod.err || { echo "rc $?"; tail -30 $OUT/bench_dvbs2_demod.err; }
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_dvbs2_demod.json"))
    for k in ("value","unit","Msamples_per_s","frames_per_s","realtime_factor_at_45_Msym_per_s","ms_per_step","bbframes_per_step","all_bbframes_are_transmitted_ones_in_order","frames_not_matching","pll_schedule_per_step","kernels_ms","roofline","whole_path","cpu_baseline","parity_sample","stats"):
        print(k, d.get(k))
except Exception as e:
    print("no result", e)
PY
echo "== driver workload, short: exact + streamed legs"
timeout 1200 python bench.py --gpus 1 --steps 5 --warmup 2 --parity-samples 200000000 --others 0 --next-rows 0 > $OUT/bench_short.json 2> $OUT/bench.err || { echo "bench rc $?"; tail -20 $OUT/bench.err; }
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_short.json"))
    print({k:d[k] for k in ("value","ms_per_step","roofline","exact_mode","streamed","parity_gates")})
    print("soft_parity", d["soft_parity"]["frac_within_1e-5"], d["soft_parity"]["max_lsb"], "cadu", d["cadu_parity"]["byte_identical"], d["cadu_parity"]["compared"])
    print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:10]))
except Exception as e:
    print("no result", e)
PY
