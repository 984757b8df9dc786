#!/usr/bin/env python3
"""Where the symbols sit that the chunk-parallel clock recovery interpolates TWO OR MORE grid steps from the reference's position (bench.py soft_parity.arm_grid.other),
on the host twin with the bench's own chunk geometry: python tools/twin/arm_probe.py goes 3712 [nframes] -- prints runs of such symbols and where in their chunk they lie."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.emu import build as emu_build, fake_torch  # noqa: E402

lib = emu_build.build()
os.environ["SDHIP_LIB"] = lib
os.environ["SDHIP_TESTING_TWIN"] = "1"
from satdump_amd import capi, synth  # noqa: E402
from oracle import pyref  # noqa: E402
from tests import util, test_demod_gpu as G  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "goes"
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 3712
nframes = int(sys.argv[3]) if len(sys.argv) > 3 else 100
if case == "goes":
    spec, cadus, plain, syms = util.goes_case(nframes=nframes)
elif case == "metop":
    spec, cadus, plain, syms = util.metop_case(nframes=nframes)
else:
    spec, cadus, plain, syms = util.npp_case(nframes=nframes)
_, _, _, ocfg, kw, _, _ = G._case(case)
x, _ = synth.modulate(syms, spec)
print(f"{case}: {len(x)} samples, chunk_len {chunk}", flush=True)
want, ref_pos = pyref.psk_demod_with_arms(ocfg, x)
soft, sy, st = G._run_demod(fake_torch, capi, kw, x, chunk_len=chunk)
_, taps, st2 = G._run_demod(fake_torch, capi, kw, x, chunk_len=chunk, tap=1)
pos = taps.view(np.int64)
n = min(len(pos), len(ref_pos))
step = pos[:n] - ref_pos[:n]
ref = want["syms"][:n]
err = np.abs(sy[:n] - ref) / np.sqrt(np.mean(np.abs(ref) ** 2))
other = np.abs(step) >= 2
print(f"chunks {st.chunks} fixed {st.chunks_fixed} inexact {st.chunks_inexact}; symbols {n}: same {np.mean(step == 0):.6f} one {np.mean(np.abs(step) == 1):.6f} other {other.sum()} "
      f"max|step| {np.abs(step).max()}; beyond 1e-5 {np.mean(err > 1e-5):.6f}; same-arm beyond 1e-5 {(err[step == 0] > 1e-5).sum()}")
amp = np.abs(np.abs(sy[:n]) - np.abs(ref)) / np.sqrt(np.mean(np.abs(ref) ** 2))
ang = np.abs(np.angle(sy[:n] * np.conj(ref)))
sb = np.flatnonzero((step == 0) & (err > 1e-5))
if len(sb):
    runs = np.split(sb, np.flatnonzero(np.diff(sb) > 400) + 1)
    print(f"same-arm beyond 1e-5: {len(sb)} symbols in {len(runs)} clusters; max angle {ang[sb].max():.3g}, max amplitude diff {amp[sb].max():.3g}")
    for r in runs[:30]:
        a, b = r[0], r[-1]
        s0 = ref_pos[a] // 128
        print(f"  symbols {a}..{b} ({len(r)}), sample {s0} = chunk {s0 / chunk:.3f}; angle first {ang[a]:.3g} max {ang[r].max():.3g}, amplitude max {amp[r].max():.3g}")
idx = np.flatnonzero(other)
if len(idx):
    runs = np.split(idx, np.flatnonzero(np.diff(idx) > 200) + 1)
    print(f"{len(runs)} clusters")
    for r in runs[:40]:
        a, b = r[0], r[-1]
        # input sample position of the symbol (resampled-rate index): ref_pos / 128
        s0 = ref_pos[a] // 128
        print(f"  symbols {a}..{b} ({len(r)} of {b - a + 1}), sample {s0} = chunk {s0 / chunk:.3f}, steps {np.unique(step[r])}, |step| profile around: "
              f"{np.abs(step[max(0, a - 3):a + 12]).tolist()}")
