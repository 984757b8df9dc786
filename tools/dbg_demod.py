import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from oracle import pyref
from satdump_amd import capi, synth
from tests import util
from tests.test_demod_gpu import _case, _run_demod
orc = pyref.best()
for case in ["goes", "metop", "npp"]:
    spec, plain, x, ocfg, kw, fec, ofec = _case(case)
    want = orc.psk_demod(ocfg, x)
    for extra in [dict(chunk_len=8192), dict(chunk_len=4096)]:
        soft, syms, st = _run_demod(torch, capi, kw, x, **extra)
        ref = want["syms"]
        scale = np.sqrt(np.mean(np.abs(ref) ** 2))
        n = min(len(ref), len(syms))
        err = np.abs(syms[:n] - ref[:n]) / scale
        bad = np.flatnonzero(err > 1e-5)
        print(case, extra, "nsym", len(syms), len(ref), "chunks", st.chunks, "fixed", st.chunks_fixed, "rot", st.chunks_rotated, "inexact", st.chunks_inexact,
              "maxerr %.3g" % err.max(), "nbad", len(bad), "first bad", bad[:5], "last bad", bad[-5:], "sps", st.final_sps)
        d = soft.astype(int) - want["soft"].astype(int); print("   int8 diffs", np.count_nonzero(d), "of", len(d), "max", np.abs(d).max())
        if len(bad):
            # histogram of bad symbol positions in units of chunk (in input samples: sym idx * sps)
            pos = bad * st.final_sps
            print("   bad sample pos/8192 (first 12):", np.round(pos[:: max(1, len(pos)//12)] / 8192, 2)[:12])
            print("   err quantiles", np.quantile(err, [0.5, 0.9, 0.99, 0.999]))
