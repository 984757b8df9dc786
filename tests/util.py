"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from satdump_amd import synth


def goes_case(nframes=40, seed=2, esn0_db=7.0, amplitude=0.5, cfo_hz=1000.0):
    """BASELINE config 2 in miniature: GOES HRIT BPSK 927 ksym/s @ 3 Msps, r=1/2, NRZ-M, RS I=4."""
    spec = synth.SynthSpec(constellation="bpsk", samplerate=3e6, symbolrate=927e3, nrzm=True, esn0_db=esn0_db, amplitude=amplitude,
                           cfo_hz=cfo_hz, seed=seed)
    cadus = synth.make_cadus(nframes, seed=seed)
    plain = synth.make_cadus(nframes, seed=seed, derand=False)
    syms = synth.frames_to_symbols(cadus, spec)
    return spec, cadus, plain, syms


def metop_case(nframes=60, seed=1, esn0_db=10.0):
    """BASELINE configs 1/3 in miniature: MetOp AHRPT QPSK 2.333 Msym/s @ 6 Msps, r=3/4 MetOp puncture."""
    spec = synth.SynthSpec(constellation="qpsk", samplerate=6e6, symbolrate=2333333, conv="3/4-metop", nrzm=False, cfo_hz=3000,
                           esn0_db=esn0_db, amplitude=0.25, seed=seed)
    cadus = synth.make_cadus(nframes, seed=seed)
    plain = synth.make_cadus(nframes, seed=seed, derand=False)
    syms = synth.frames_to_symbols(cadus, spec)
    return spec, cadus, plain, syms


def npp_case(nframes=60, seed=4, esn0_db=8.0):
    """BASELINE config 4 in miniature: JPSS HRD QPSK 15 Msym/s @ 30 Msps, r=1/2, NRZ-M."""
    spec = synth.SynthSpec(constellation="qpsk", samplerate=30e6, symbolrate=15e6, conv="1/2", nrzm=True, cfo_hz=20000,
                           esn0_db=esn0_db, amplitude=0.4, seed=seed)
    cadus = synth.make_cadus(nframes, seed=seed)
    plain = synth.make_cadus(nframes, seed=seed, derand=False)
    syms = synth.frames_to_symbols(cadus, spec)
    return spec, cadus, plain, syms


def frame_ids(out, plain):
    """Index of each decoded CADU in the transmitted (derandomised) list, -1 if none matches."""
    ids = []
    for o in out:
        m = np.flatnonzero((plain == o).all(1))
        ids.append(int(m[0]) if len(m) else -1)
    return ids
