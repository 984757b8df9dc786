#!/bin/bash
# GPU visit h: parity suite, then the back-to-back and the overlapped (two host threads, two streams) bench lines of every workload.
TAG=${1:-r02_h}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
run() { # name, args...
  local name=$1; shift
  SDHIP_DEBUG=1 timeout 900 python bench.py --cpu-samples 0 "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo FAILED $name; tail -5 $OUT/$name.err; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json"))
print("$name", d["value"], d["ms_per_step"], d["check"]["cadus_matching_transmitted"] if d.get("check") else None)
print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}/{v['launches_per_step']:.0f}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]))
PY
  grep -E "search\+pack|frames\+rs|walk" $OUT/$name.err | tail -6
}
run metop --steps 4
run metop_pipe --steps 6 --pipeline
run npp --workload npp_hrd --steps 4
run npp_pipe --workload npp_hrd --steps 6 --pipeline
run goes --workload goes_hrit --steps 6
run goes_pipe --workload goes_hrit --steps 8 --pipeline
