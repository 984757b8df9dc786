#!/bin/bash
TAG=${1:-r02_n}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
run() { # name, args
  local name=$1; shift
  env $ENVV SDHIP_DEBUG=1 timeout 400 python bench.py --steps 4 --warmup 1 --cpu-samples 0 "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo FAILED $name; tail -3 $OUT/$name.err; return; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json"))
print("$name", d["value"], d["ms_per_step"], d["check"]["cadus_matching_transmitted"] if d.get("check") else None)
print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}/{v['launches_per_step']:.0f}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:13]))
PY
  grep -E "walk:|off the gathered" $OUT/$name.err | tail -3
}
ENVV="X=1" run metop
ENVV="X=1" run npp --workload npp_hrd
ENVV="X=1" run goes --workload goes_hrit
