#!/bin/bash
TAG=${1:-r02_l}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/ubench/lane_layout.hip -o /tmp/lane_layout && { /tmp/lane_layout 65280 32768; /tmp/lane_layout 196608 10880; /tmp/lane_layout 16320 131072; } | tee $OUT/lane_layout.txt
timeout 600 python -m pytest tests/test_demod_gpu.py tests/test_zy_demod_additions_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_demod.txt
run() { # name, args
  local name=$1; shift
  env $ENVV timeout 300 python bench.py --cpu-samples 0 --steps 3 --warmup 1 "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo FAILED $name; tail -3 $OUT/$name.err; return; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json"))
print("$name", d["value"], d["ms_per_step"], d["check"]["cadus_matching_transmitted"] if d.get("check") else None)
print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}/{v['launches_per_step']:.0f}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:13]))
PY
}
ENVV="X=1" run metop
ENVV="SDHIP_MM_Q8=0" run metop_noq8
ENVV="X=1" run goes --workload goes_hrit
ENVV="X=1" run npp --workload npp_hrd
