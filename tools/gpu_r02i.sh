#!/bin/bash
# GPU visit i: parity suite + the bench lines of every workload (no CPU legs)
TAG=${1:-r02_i}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
run() { # name, args...
  local name=$1; shift
  SDHIP_DEBUG=1 timeout 900 python bench.py --cpu-samples 0 "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo FAILED $name; tail -5 $OUT/$name.err; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json"))
print("$name", d["value"], d["ms_per_step"], d["check"]["cadus_matching_transmitted"] if d.get("check") else None)
print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}/{v['launches_per_step']:.0f}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]))
PY
}
run metop --steps 4
run npp --workload npp_hrd --steps 4
run goes --workload goes_hrit --steps 6
