"""Differential fuzz of the DVB-S2 synchroniser / PLL / demapper entries on the HOST TWIN against the reference's own blocks (oracle/_ref): random MODCOD
(QPSK / 8PSK / 16APSK, short frames), pilots, lead-in noise, slipped symbols, carrier offset, noise level, table-range overshoot; every entry must be
bit-identical.  usage: python tools/twin/dvbs2_fuzz.py [seed]"""
import ctypes as C
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import pyref
from tests import dvbs2_util
from tests.emu import build as emu_build


def main():
    lib = emu_build.build()
    os.environ["SDHIP_LIB"] = lib
    os.environ["SDHIP_TESTING_TWIN"] = "1"
    spec = importlib.util.spec_from_file_location("capi_tw", os.path.join(ROOT, "satdump_amd", "capi.py"))
    capi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(capi)
    L = capi.lib()
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    front = pyref.S2FrontRef()
    bad = 0
    for t in range(16):
        modcod = int(rng.choice([1, 4, 6, 11, 12, 13, 17, 18, 20, 23]))
        short, pilots = 1, int(rng.integers(0, 2))
        c = front.cfg(modcod, short, pilots)
        probe = np.zeros(8, dtype=np.complex64)
        _, _, raw = pyref.s2_pl_sync_ref(c["slots"], pilots, 0.6, probe, max_frames=1)
        nfr = int(rng.integers(3, 8))
        gl = {int(f): int(rng.integers(1, raw)) for f in rng.choice(nfr, int(rng.integers(0, 3)), replace=False)}
        x = dvbs2_util.pl_stream(raw, (modcod << 2) | (short << 1) | pilots, nfr, seed=int(rng.integers(1, 10 ** 6)), lead=int(rng.integers(0, raw)), glitches=gl,
                                 esn0_db=float(rng.uniform(2, 16)), cfo=float(rng.uniform(-0.003, 0.003)), amplitude=float(rng.uniform(0.2, 1.3)))
        thr = float(rng.choice([0.6, 0.3, 0.9]))
        want, wcons, _ = pyref.s2_pl_sync_ref(c["slots"], pilots, thr, x)
        cap = len(x) // raw + 2
        stride = raw + int(rng.integers(0, 9))
        d_fr = np.zeros((cap, stride), dtype=np.complex64)
        bp = np.full(cap, -1, dtype=np.int32)
        consumed = C.c_size_t(0)
        nf = L.sdhip_s2_pl_sync_dev(0, c["slots"], pilots, thr, x.ctypes.data_as(C.c_void_p), len(x), d_fr.ctypes.data_as(C.c_void_p), stride, cap, C.byref(consumed),
                                    bp.ctypes.data_as(C.c_void_p))
        ok = nf >= len(want) and np.array_equal(bp[:len(want)] + raw, wcons) and np.array_equal(d_fr[:len(want), :raw].view(np.uint32), want.view(np.uint32))
        nf = len(want)
        if nf and ok:
            bw = float(rng.choice([0.002, 0.01, 0.0005]))
            wp, walked, wst = pyref.s2_pll_ref(modcod, short, pilots, bw, want)
            lutp, lutb = pyref.s2_lut_phase_ref(modcod, short), front.lut(modcod, short)
            d_pl = np.zeros_like(d_fr)
            st = np.zeros(2, dtype=np.float32)
            r = L.sdhip_s2_pll_dev(0, modcod, short, pilots, bw, d_fr.ctypes.data_as(C.c_void_p), d_pl.ctypes.data_as(C.c_void_p), stride, nf, lutp.ctypes.data_as(C.c_void_p), 256,
                                   st.ctypes.data_as(C.c_void_p))
            ok = ok and r == walked and np.array_equal(d_pl[:nf, :walked].view(np.uint32), wp[:, :walked].view(np.uint32)) and np.array_equal(st.view(np.uint32), wst.view(np.uint32))
            # the demapper stage reads whole frames: hand both sides the same complete rows (the PLL leaves the tail of a pilots frame unwritten)
            rows = wp.copy()
            rows[:, walked:] = want[:, walked:]
            ws, wpls = front.bb_to_soft(modcod, short, pilots, rows)
            d_rows = np.zeros((nf, stride), dtype=np.complex64)
            d_rows[:, :raw] = rows
            nsoft = c["slots"] * 90 * c["bits"]
            d_soft = np.zeros((nf, nsoft), dtype=np.int8)
            d_pls = np.zeros(nf, dtype=np.int32)
            r = L.sdhip_s2_bb_to_soft_dev(0, modcod, short, pilots, d_rows.ctypes.data_as(C.c_void_p), stride, nf, lutb.ctypes.data_as(C.c_void_p), 256, d_soft.ctypes.data_as(C.c_void_p),
                                          d_pls.ctypes.data_as(C.c_void_p))
            ok = ok and r == nsoft and np.array_equal(d_soft, ws) and np.array_equal(d_pls, wpls)
        if not ok:
            bad += 1
            print("MISMATCH", t, dict(modcod=modcod, pilots=pilots, nfr=nfr, glitches=gl, thr=thr), capi.last_error())
    print("trials done, mismatches:", bad)


if __name__ == "__main__":
    main()
