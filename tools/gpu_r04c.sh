#!/bin/bash
# Round 4, visit C: multi-rank flow with aligned decoders (whole-frame identity with the single stream), the plugin's hip_devices, host-path rates
TAG=${1:-r04_c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_multirank_gpu.py tests/test_plugin_minihost_gpu.py tests/test_dvbs2_gpu.py -m gpu -q -k "two_ranks or hip_devices or pll_parallel or stock_ids or dvbs2_module" 2>&1 | tail -12 | tee $OUT/pytest_new.txt
echo "== host copy ubench"
hipcc -O2 -o /tmp/host_copy tools/ubench/host_copy.hip -lpthread 2>&1 | tail -2; /tmp/host_copy | tee $OUT/host_copy.txt
echo "== streamed leg (driver workload, no reference legs)"
for t in 8 16 32; do
SDHIP_COPY_THREADS=$t timeout 900 python - <<PY 2>&1 | tail -3 | tee -a $OUT/streamed.txt
import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from satdump_amd import capi
wl = bench.WORKLOADS["metop_ahrpt"]
from satdump_amd import synth
rec = synth.Recording(synth.SynthSpec(**wl["spec"]), wl["frames"], blocks=1)
nst = 1 << 29
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
    print(os.environ.get("SDHIP_COPY_THREADS"), kind, [round(nst * 8 / t / 1e9, 2) for t in times], "GB/s", nsoft)
PY
done
