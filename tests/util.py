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


# ---------------------------------------------------------------- ccsds_simple_psk_decoder inputs (uncoded PSK)
def _qpsk_diff_decode(prev, cur):
    """diff::QPSKDiff's rule for one symbol pair (qpsk_diff.cpp:27-41); symbols are 2*(Q>0) + (I>0)."""
    xin_1, yin_1, xin, yin = prev & 2, prev & 1, cur & 2, cur & 1
    if ((xin >> 1) ^ yin) == 1:
        xout, yout = yin_1 ^ yin, xin_1 ^ xin
        return (xout << 1) + (yout >> 1)
    xout, yout = xin_1 ^ xin, yin_1 ^ yin
    return xout + yout


def simple_soft(cadus, constellation="bpsk", nrzm=False, sigma=20.0, seed=0, rot90=False, swap_diff=True, amp=None):
    """int8 .soft stream of uncoded CADUs as psk_demod would write it (BPSK 1 B/symbol, QPSK I,Q interleaved), such that
    ccsds_simple_psk_decoder with the matching options recovers the frames. rot90: send the QPSK stream a quarter turn
    off, so that the decoder's second (rotated) deframer is the one that locks."""
    rng = np.random.default_rng(seed)
    bits = np.unpackbits(np.ascontiguousarray(cadus).reshape(-1)).astype(np.int64)
    if constellation == "bpsk":
        if nrzm:
            bits = synth.nrzm_encode(bits.astype(np.uint8)).astype(np.int64)
        a = 50.0 if amp is None else amp
        v = (2 * bits - 1) * a + sigma * rng.standard_normal(len(bits))
    else:
        a = 70.0 if amp is None else amp
        pairs = bits.reshape(-1, 2)
        if nrzm:
            # differential encoder matched to diff::QPSKDiff by search: out pair (o0, o1) <- symbol pair (prev, cur)
            syms = np.zeros(len(pairs) + 2, dtype=np.int64)
            prev = 0
            for k, (o0, o1) in enumerate(pairs):
                ou = (o0 | (o1 << 1)) if swap_diff else ((o0 << 1) | o1)
                cur = next(c for c in range(4) if _qpsk_diff_decode(prev, c) == ou)
                syms[k + 2] = cur
                prev = cur
            q, i = syms >> 1, syms & 1
        else:
            q, i = pairs[:, 0], pairs[:, 1]  # bits_out[2k] = sym >> 1 = (Q > 0), bits_out[2k+1] = sym & 1 = (I > 0)
        I, Q = (2 * i - 1) * a, (2 * q - 1) * a
        if rot90:  # the decoder applies PHASE_90: (I, Q) -> (Q, -I); pre-rotate by the inverse
            I, Q = -Q, I
        v = np.empty(2 * len(I))
        v[0::2], v[1::2] = I, Q
        v = v + sigma * rng.standard_normal(len(v))
    return np.where(v < -128.0, -127, np.where(v > 127.0, 127, np.trunc(v))).astype(np.int8)


# (name, oracle/capi cfg keywords, synth keywords) -- the option space of ccsds_simple_psk_decoder
SIMPLE_CASES = [
    ("bpsk", dict(constellation="bpsk", nrzm=0), dict(constellation="bpsk", nrzm=False)),
    ("bpsk_nrzm", dict(constellation="bpsk", nrzm=1), dict(constellation="bpsk", nrzm=True)),
    ("qpsk_0deg", dict(constellation="qpsk", nrzm=0), dict(constellation="qpsk")),
    ("qpsk_90deg", dict(constellation="qpsk", nrzm=0), dict(constellation="qpsk", rot90=True)),
    ("qpsk_diff_swap", dict(constellation="qpsk", nrzm=1, qpsk_swap_diff=1), dict(constellation="qpsk", nrzm=True, swap_diff=True)),
    ("qpsk_diff_noswap", dict(constellation="qpsk", nrzm=1, qpsk_swap_diff=0), dict(constellation="qpsk", nrzm=True, swap_diff=False)),
    ("qpsk_method2", dict(constellation="qpsk", nrzm=0, oqpsk_method2=1), dict(constellation="qpsk")),
    ("qpsk_method3", dict(constellation="qpsk", nrzm=0, oqpsk_method3=1), dict(constellation="qpsk")),
    ("qpsk_swapiq", dict(constellation="qpsk", nrzm=0, qpsk_swap_iq=1), dict(constellation="qpsk", rot90=True)),
    ("qpsk_delay_diff", dict(constellation="qpsk", nrzm=1, oqpsk_delay=1, qpsk_swap_iq=1), dict(constellation="qpsk", nrzm=True)),
]


def simple_case(name, sigma=15.0, nframes=12, seed=5, prefix=3000):
    """(cfg keywords, soft stream, transmitted derandomised frames) of one SIMPLE_CASES entry, with a junk prefix in front."""
    _, ck, sk = next(c for c in SIMPLE_CASES if c[0] == name)
    cadus = synth.make_cadus(nframes, seed=seed)
    plain = synth.make_cadus(nframes, seed=seed, derand=False)
    soft = simple_soft(cadus, sigma=sigma, seed=seed + 1, **sk)
    junk = np.random.default_rng(seed + 2).integers(-90, 90, prefix).astype(np.int8)
    return dict(ck), np.concatenate([junk, soft]), plain


# ---------------------------------------------------------------- conv_rate != 1/2 (Viterbi_Depunc) inputs
def punctured_case(rate, nframes=10, sigma=20.0, seed=3, prefix=777, nrzm=False, gap=False):
    """int8 soft stream of a punctured (rate code 1 = 2/3, 2 = 3/4, 3 = 5/6, 4 = 7/8) r=1/2 k=7 stream carrying `nframes` CADUs,
    behind `prefix` garbage symbols; gap=True inserts 5 blocks of noise in the middle (loss of lock + re-lock at another shift).
    Returns (soft, plain frames as the decoder should deliver them)."""
    from satdump_amd import synth
    cadus = synth.make_cadus(nframes, seed=seed)
    plain = synth.make_cadus(nframes, seed=seed, derand=False)
    bits = np.unpackbits(cadus.reshape(-1))
    if nrzm:
        bits = synth.nrzm_encode(bits)
    tx = synth.puncture(synth.conv_encode(bits), rate)
    rng = np.random.default_rng(seed + 100 * rate)
    soft = np.clip(np.rint((tx.astype(float) * 2 - 1) * 60 + rng.standard_normal(len(tx)) * sigma), -127, 127).astype(np.int8)
    if gap:
        h = len(soft) // 2 + 3
        soft = np.concatenate([soft[:h], rng.integers(-127, 128, 5 * 8192 + 1).astype(np.int8), soft[h:]])
    soft = np.concatenate([rng.integers(-127, 128, prefix).astype(np.int8), soft])
    pad = (-len(soft)) % 8192
    soft = np.concatenate([soft, rng.integers(-127, 128, pad + 8192).astype(np.int8)])
    return soft, plain
