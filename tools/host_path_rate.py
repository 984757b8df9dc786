"""PCIe-inclusive rate of the host-buffer entry points (what the C++ modules call): cf32 samples in pageable host memory ->
sdhip_demod_push/flush/pull -> sdhip_fec_push/pull -> CADUs in host memory. Never the bench's `value`."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from satdump_amd import capi

wl = bench.WORKLOADS["goes_hrit"]
x, plain, spec = bench.make_input(wl, torch.device("cuda", 0), 0, wl["frames"])
xh = x.cpu().numpy()  # pageable
n = len(xh)
dem = capi.PskDemod(capi.demod_cfg(**wl["demod"]))
fec = capi.FecDecoder(capi.fec_cfg(**wl["fec"]))
for it in range(3):
    t0 = time.perf_counter()
    step = 16 << 20
    for a in range(0, n, step):
        dem.push(xh[a:a + step])
    dem.flush()
    soft = dem.pull(2 * n)
    t1 = time.perf_counter()
    fec.push(soft)
    cadu = fec.pull()
    t2 = time.perf_counter()
    print(f"pass {it}: {n} samples, demod host path {t1 - t0:.3f} s, fec host path {t2 - t1:.3f} s -> {n / (t2 - t0) / 1e6:.0f} Msamples/s, {len(cadu)} CADUs")
