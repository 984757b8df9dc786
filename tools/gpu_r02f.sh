#!/bin/bash
TAG=${1:-r02_f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
run() { # name, env...
  local name=$1; shift
  env "$@" SDHIP_DEBUG=1 timeout 900 python bench.py --cpu-samples 0 > $OUT/$name.json 2> $OUT/$name.err || { echo FAILED $name; tail -5 $OUT/$name.err; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json"))
print("$name", d["value"], d["ms_per_step"], d["check"]["cadus_matching_transmitted"])
print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}/{v['launches_per_step']:.0f}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]))
PY
  grep -E "search\+pack|frames\+rs" $OUT/$name.err | tail -4
}
run base
run norerun SDHIP_MM_TOL_MICRO=80000 SDHIP_COSTAS_TOL_URAD=60000 SDHIP_COSTAS_TOL_NFREQ=200000
run nockpt SDHIP_CKPT=0 SDHIP_MM_TOL_MICRO=80000 SDHIP_COSTAS_TOL_URAD=60000 SDHIP_COSTAS_TOL_NFREQ=200000
run mm48k SDHIP_LANES_MM=49152
run cos262k SDHIP_LANES_COSTAS=262144
