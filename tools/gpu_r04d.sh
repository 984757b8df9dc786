#!/bin/bash
# Round 4, visit D: multi-rank whole-frame identity (full failure output), hip_devices, ndsp single blocks, host path with the two-thread pipeline
TAG=${1:-r04_d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -q -x 2>&1 | tail -60 | tee $OUT/pytest_multirank.txt
timeout 900 python -m pytest tests/test_plugin_minihost_gpu.py tests/test_ndsp_gpu.py tests/test_dvbs2_gpu.py -m gpu -q -k "hip_devices or single_block or pll_parallel or ndsp_block or stock_ids" 2>&1 | tail -30 | tee $OUT/pytest_new.txt
echo "== streamed"
timeout 900 python - <<PY 2>&1 | tail -4 | tee $OUT/streamed.txt
import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from satdump_amd import capi, synth
wl = bench.WORKLOADS["metop_ahrpt"]
rec = synth.Recording(synth.SynthSpec(**wl["spec"]), wl["frames"], blocks=1)
nst = 1 << 30
x = rec.synth_range(0, nst, device=torch.device("cuda", 0))
xs = x.cpu().numpy()
del x
sink = np.empty(64 << 20, dtype=np.int8)
for kind in ("pageable", "pinned"):
    src = xs if kind == "pageable" else torch.from_numpy(xs).pin_memory().numpy()
    ds = capi.PskDemod(capi.demod_cfg(**wl["demod"]))
    times = []
    for rep in range(3):
        ts = time.perf_counter(); nsoft = 0
        for a0 in range(0, nst, 4 << 20):
            ds.push(src[a0:a0 + (4 << 20)])
            while True:
                g = ds.pull(out=sink); nsoft += len(g)
                if len(g) < sink.size: break
        ds.flush()
        while True:
            g = ds.pull(out=sink); nsoft += len(g)
            if len(g) < sink.size: break
        times.append(time.perf_counter() - ts)
    ds.close()
    print(kind, [round(nst * 8 / t / 1e9, 2) for t in times], "GB/s", nsoft)
PY
