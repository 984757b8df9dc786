#!/usr/bin/env python3
"""The reference's DVB-S2 demodulator chain on ONE host thread (its blocks and classes chained the way DVBS2DemodModule chains them: AGC, RRC filter, M&M,
S2PLSyncBlock, S2PLLBlock, S2BBToSoft, BBFrameLDPC, BBFrameBCH, BB descrambler -- oracle/_ref), as a function for tools/bench_dvbs2_demod.py and as a worker
process of its all-cores leg:   dvbs2_cpu_chain.py <samples.npy> <modcod> <symrate> <sps> <alpha> <loop_bw> <trials> <batch>   prints the seconds it took."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def chain(xs, modcod, symrate, sps, alpha, loop_bw, trials, batch):
    """-> (BBFRAMEs, symbols behind the clock recovery, stage seconds, LDPC trial counts)"""
    from oracle import pyref
    from satdump_amd import synth_dvbs2 as sd
    c = sd.modcod_cfg(modcod, 0)
    orc = pyref.best()
    fref = pyref.S2FrontRef()
    fec = pyref.Dvbs2Ref(pyref.Dvbs2Ref.available(True) and batch == 16)
    rc = c["rate"]
    nl, kl = fec.dims(0, rc)
    kb = fec.bch_kbch(0, rc)
    t1 = time.perf_counter()
    xr = orc.block(3, [float(sps), (1.7e-3) ** 2 / 4, 0.5, 1.7e-3, 0.005], orc.block(1, [symrate * sps, symrate, alpha, 31], orc.block(0, [1e-2, 1.0, 1.0, 65536.0], xs)))
    t2 = time.perf_counter()
    fr, _, _ = pyref.s2_pl_sync_ref(c["slots"], 0, 0.6, xr)
    rp, _, _ = pyref.s2_pll_ref(modcod, 0, 0, loop_bw, fr)
    soft, _ = fref.bb_to_soft(modcod, 0, 0, rp)
    nfull = len(soft) // fec.batch * fec.batch
    t3 = time.perf_counter()
    dec, tr = fec.ldpc_decode(0, rc, soft[:nfull].copy(), trials)
    t4 = time.perf_counter()
    fix, _ = fec.bch_decode(0, rc, np.packbits((dec < 0).astype(np.uint8), axis=1)[:, :kl // 8].copy())
    want = fec.bb_descramble(0, rc, fix.copy())[:, :kb // 8]
    t5 = time.perf_counter()
    return want, len(xr), {"front_end": round(t2 - t1, 3), "sync_pll_demap": round(t3 - t2, 3), "ldpc": round(t4 - t3, 3), "bch_descramble": round(t5 - t4, 3), "total": t5 - t1}, tr, fec.batch


if __name__ == "__main__":
    a = sys.argv
    x = np.load(a[1])
    _, nsym, st, _, _ = chain(x, int(a[2]), float(a[3]), int(a[4]), float(a[5]), float(a[6]), int(a[7]), int(a[8]))
    print(st["total"], nsym, flush=True)
