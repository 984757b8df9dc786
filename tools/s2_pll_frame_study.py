#!/usr/bin/env python3
"""How a frame-parallel schedule of the DVB-S2 frame PLL would have to be set up, measured on the REFERENCE's own loop (oracle/_ref,
S2PLLBlock started from a given state): every frame is run as a lane that starts at its own header from (a) phase 0 and the stream's mean frequency,
(b) the serial loop's frequency and a data-aided phase estimate from the 90 known header symbols; the lane's symbols are compared with the serial
loop's, as a function of how far into the frame one looks.   usage: python tools/s2_pll_frame_study.py [modcod] [short] [esn0_db] [cfo]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import pyref
from tests import dvbs2_util


def pll_from(lib, modcod, short, bw, frames, state):
    f = np.ascontiguousarray(frames, dtype=np.complex64)
    out = np.zeros_like(f)
    st = np.array(state, dtype=np.float32)
    lib.sdref_s2_pll_from.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    w = lib.sdref_s2_pll_from(modcod, short, 0, bw, f.ctypes.data_as(C.c_void_p), f.shape[1], len(f), out.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p))
    return out[:, :w], st


def main():
    modcod = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    short = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    esn0 = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
    cfo = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0004
    bw = 0.002
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsdref_dvbs2.so"))
    front = pyref.S2FrontRef()
    c = front.cfg(modcod, short, 0)
    raw = (c["slots"] + 1) * 90
    nfr = 12
    x = dvbs2_util.pl_stream(raw, (modcod << 2) | (short << 1), nfr + 1, seed=7, lead=0, cfo=cfo, esn0_db=esn0, amplitude=2.0 / 3.0)
    fr = x[: nfr * raw].reshape(nfr, raw)
    # the serial loop: state in front of every frame
    states = [np.zeros(2, dtype=np.float32)]
    serial = []
    for f in range(nfr):
        o, st = pll_from(lib, modcod, short, bw, fr[f:f + 1], states[-1])
        serial.append(o[0])
        states.append(st)
    serial = np.array(serial)
    hdr = np.concatenate([dvbs2_util.sof_symbols(), dvbs2_util.pls_symbols((modcod << 2) | (short << 1))])
    print(f"modcod {modcod} {'short' if short else 'normal'} frames ({raw} symbols), Es/N0 {esn0} dB, offset {cfo} rad/symbol, loop_bw {bw}; serial loop frequency {states[-1][1]:.6f}")
    for name in ("phase 0, serial frequency", "header estimate, serial frequency", "header estimate, frequency from two headers"):
        rows = []
        for f in range(4, nfr):
            fs = float(states[f][1])
            if name.startswith("phase 0"):
                st0 = [0.0, fs]
            else:
                # data-aided: the rotation that turns the received header onto the known one, at the header's centre, walked back to symbol 0
                z = np.vdot(hdr, fr[f, :90])            # sum conj(known) * received
                ph_c = np.angle(z)
                if name.endswith("two headers"):
                    zp = np.vdot(hdr, fr[f - 1, :90])
                    d = np.angle(z * np.conj(zp))
                    k = round((fs * raw - d) / (2 * np.pi))       # the serial frequency only picks the branch
                    fs = (d + 2 * np.pi * k) / raw
                st0 = [float(ph_c - fs * 44.5), fs]
            o, _ = pll_from(lib, modcod, short, bw, fr[f:f + 1], st0)
            err = np.abs(o[0] - serial[f]) / np.sqrt(np.mean(np.abs(serial[f]) ** 2))
            rows.append([float(np.max(err[a:b])) for a, b in ((90, 400), (400, 1000), (1000, 2000), (2000, 4000), (4000, 8000), (8000, raw))])
        r = np.array(rows)
        print(f"  lane start = {name:45s} max rel. symbol error vs the serial loop, by symbol range [90,400) [400,1k) [1k,2k) [2k,4k) [4k,8k) [8k,end): "
              + "  ".join(f"{v:.1e}" for v in r.max(axis=0)))


if __name__ == "__main__":
    main()
