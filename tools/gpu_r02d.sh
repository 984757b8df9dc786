#!/bin/bash
# parity suite (no -x: every failure shown), MetOp + GOES full benches, SQ counters on GOES
TAG=${1:-r02_d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
for WL in metop_ahrpt goes_hrit npp_hrd; do
  echo "== full $WL"; SDHIP_DEBUG=1 timeout 900 python bench.py --workload $WL $([ $WL = npp_hrd ] && echo --cpu-samples 0) > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err || { echo FAILED; tail -30 $OUT/bench_$WL.err; }
  grep -E "re-run|early|warm-up" $OUT/bench_$WL.err | tail -6
  python - <<PY
import json
d=json.load(open("$OUT/bench_$WL.json"))
print({k:d[k] for k in ("value","ms_per_step","soft_parity","cadu_parity","check")})
print(d["roofline"]); print(d["cpu_baseline"])
print(' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}/{v['launches_per_step']:.0f}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:14]))
PY
done
tools/gpu_sq.sh $TAG goes_hrit 2>&1 | tail -30
