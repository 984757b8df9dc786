#!/bin/bash
# Round 3, visit A: the fused AGC + filter + Costas stage. Parity suite (without the two 17 GB reference decodes -- the bench line does
# the MetOp one), A/B of the new stage's switches, the driver line with its full-stream parity, kernel stats.
TAG=${1:-r03_a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "not full_size_metop and not full_size_npp" 2>&1 | tail -15 > $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
timeout 600 python tools/ab_demod.py --workload metop_ahrpt "" "SDHIP_LANES_AFC=130560" "SDHIP_COSTAS_TAUS=16" "SDHIP_LANES_AFC=130560,SDHIP_COSTAS_TAUS=16" "SDHIP_COSTAS_TAUS=12" "SDHIP_FUSE_COSTAS=0" "SDHIP_LANES_AFC=98304" > $OUT/ab_metop.txt 2> $OUT/ab_metop.err; cat $OUT/ab_metop.txt; tail -3 $OUT/ab_metop.err
timeout 300 python tools/ab_demod.py --workload goes_hrit "" "SDHIP_LANES_AFC=130560" "SDHIP_FUSE_COSTAS=0" > $OUT/ab_goes.txt 2> $OUT/ab_goes.err; cat $OUT/ab_goes.txt
echo "== driver line"; timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench rc $?"; tail -20 $OUT/bench.err; }
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","roofline")})
print("soft_parity", d["soft_parity"]); print("cadu_parity", d["cadu_parity"]); print("cpu", d["cpu_baseline"])
print("  "+' '.join(f"{n.replace('k_chunks<','').replace('Stage>','').replace('k_','')}={v['ms_per_step']}" for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]))
for k,v in d.get("other_workloads",{}).items(): print(k, {a:v[a] for a in ("value","ms_per_step","cadu_parity","top_kernels_ms")}, v["soft_parity"]["frac_within_1e-5"])
PY
