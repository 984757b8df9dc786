#!/bin/bash
# Round 5, visit P (final sources): the driver's step counts (--steps 20 --warmup 5) on the timed leg alone, twice (two processes)
TAG=${1:-r05_p}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
LEGS="--cpu-samples 0 --others 0 --next-rows 0 --exact-samples 0 --streamed-samples 0"
for i in 1 2; do timeout 110 python bench.py --gpus 1 --steps 20 --warmup 5 $LEGS > $OUT/bench_20steps_$i.json 2> $OUT/bench_20steps_$i.err || tail -5 $OUT/bench_20steps_$i.err; done
python - <<PY
import json
for i in (1, 2):
    try:
        d = json.loads(open("$OUT/bench_20steps_%d.json" % i).read().strip().splitlines()[-1])
        print(i, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], {k: v["ms_per_step"] for k, v in d["kernels"].items() if k in ("k_afc", "k_mm", "k_compact8")}, d["check"])
    except Exception as e:
        print(i, "unreadable", e)
PY
