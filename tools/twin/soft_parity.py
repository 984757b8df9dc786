"""Soft-symbol parity of the chunk-parallel demodulator against the sequential reference, measured on the HOST TWIN
(tests/emu: the unchanged HIP sources on a stand-in runtime; float arithmetic is IEEE on both sides, so fractions measured
here are the GPU's for the same chunk geometry). Usage:
    python tools/twin/soft_parity.py <goes|metop|npp> <nframes> [chunk_len] [ENV=VALUE ...]
Prints the fraction of float symbols within 1e-5 relative of the reference's, the int8 agreement and the engine's stats."""
import ctypes as C
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import pyref
from satdump_amd import synth
from tests import util
from tests.emu import build as emu_build


def load_twin():
    lib = emu_build.build()
    spec = importlib.util.spec_from_file_location("capi_host_twin", os.path.join(ROOT, "satdump_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    os.environ["SDHIP_LIB"] = lib
    os.environ["SDHIP_TESTING_TWIN"] = "1"
    spec.loader.exec_module(m)
    m.lib()
    return m


def case(name, nframes):
    if name == "goes":
        spec, cadus, plain, syms = util.goes_case(nframes=nframes)
        ocfg = pyref.demod_cfg(samplerate=3e6, symbolrate=927000, constellation=pyref.BPSK, pll_bw=0.02, max_sps=3.0)
        kw = dict(samplerate=3e6, symbolrate=927000, constellation="bpsk", rrc_alpha=0.5, pll_bw=0.02, max_sps=3.0)
    elif name == "metop":
        spec, cadus, plain, syms = util.metop_case(nframes=nframes)
        ocfg = pyref.demod_cfg(samplerate=6e6, symbolrate=2333333, constellation=pyref.QPSK, pll_bw=0.003)
        kw = dict(samplerate=6e6, symbolrate=2333333, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.003)
    else:
        spec, cadus, plain, syms = util.npp_case(nframes=nframes)
        ocfg = pyref.demod_cfg(samplerate=30e6, symbolrate=15e6, constellation=pyref.QPSK, pll_bw=0.002)
        kw = dict(samplerate=30e6, symbolrate=15e6, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.002)
    x, _ = synth.modulate(syms, spec)
    return x, ocfg, kw


def measure(syms, soft, want):
    ref = want["syms"]
    n = min(len(ref), len(syms))
    scale = np.sqrt(np.mean(np.abs(ref[:n]) ** 2))
    err = np.abs(syms[:n] - ref[:n]) / scale
    d = soft[: len(want["soft"])].astype(np.int32) - want["soft"][: len(soft)].astype(np.int32)
    return dict(n_syms=len(syms), n_ref=len(ref), frac_within_1e5=float(np.mean(err <= 1e-5)), frac_bitwise=float(np.mean(err == 0)),
                median_rel=float(np.median(err)), max_rel=float(err.max()), frac_int8_equal=float(np.mean(d == 0)), max_lsb=int(np.abs(d).max()))


def main():
    name, nframes = sys.argv[1], int(sys.argv[2])
    rest = sys.argv[3:]
    chunk_len = 0
    for a in rest:
        if "=" in a:
            k, v = a.split("=", 1)
            os.environ[k] = v
        else:
            chunk_len = int(a)
    x, ocfg, kw = case(name, nframes)
    orc = pyref.best()
    t0 = time.time()
    want = orc.psk_demod(ocfg, x)
    t1 = time.time()
    twin = load_twin()
    cfg = twin.demod_cfg(**kw, chunk_len=chunk_len)
    dem = twin.PskDemod(cfg)
    n = len(x)
    soft = np.zeros(2 * n + 64, dtype=np.int8)
    syms = np.zeros(2 * (n + 64), dtype=np.float32)
    ns = dem.process_dev(x.ctypes.data, n, twin.FMT_CF32, soft.ctypes.data, 2 * n + 64, syms.ctypes.data, n + 64)
    t2 = time.time()
    nsym = ns if cfg.constellation == twin.BPSK else ns // 2
    st = dem.stats()
    r = measure(syms[: 2 * nsym].view(np.complex64), soft[:ns], want)
    r.update(samples=n, chunks=st.chunks, fixed=st.chunks_fixed, inexact=st.chunks_inexact, forced=st.chunks_forced, t_ref=round(t1 - t0, 1), t_twin=round(t2 - t1, 1))
    print(name, nframes, chunk_len, " ".join(a for a in rest if "=" in a), r, flush=True)


if __name__ == "__main__":
    main()
