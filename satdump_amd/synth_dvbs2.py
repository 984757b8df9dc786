"""Synthetic DVB-S2 transmit side for tests and bench.py (BASELINE.json configs[4]: 8PSK rate 2/3, normal FECFRAMEs): BBFRAME bytes -> BB scrambler ->
BCH -> LDPC -> bit interleaver -> constellation mapping -> PL header + PL scrambling -> PLFRAME symbols, following ETSI EN 302 307-1 sections 5.2 - 5.5
and mapping bits to points the way the reference's own receive tables read them (so that what is made here decodes on the reference:
plugins/dvb_support/dvbs2, tools/dvbs2_mod/main.cpp is the reference's modulator and was read for the conventions -- inverted bits MSB first into
constellation_t::mod, common/dsp/demod/constellation.cpp:20-66). Pulse shaping to baseband is synth.modulate / synth.modulate_torch.

Plain numpy, no device, no oracle: a workload generator, never on a timed path. tests/ pin every stage against the compiled reference
(tests/test_synth_dvbs2_cpu.py)."""
from __future__ import annotations

import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_cache: dict = {}

RATES = {"1/4": 0, "1/3": 1, "2/5": 2, "1/2": 3, "3/5": 4, "2/3": 5, "3/4": 6, "4/5": 7, "5/6": 8, "7/8": 9, "8/9": 10, "9/10": 11}  # dvbs2_code_rate_t
_QPSK = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11]   # MODCOD 1..11
_PSK8 = [4, 5, 6, 8, 10, 11]                  # MODCOD 12..17
_APSK16 = [5, 6, 7, 8, 10, 11]                # MODCOD 18..23


def modcod_cfg(modcod: int, shortframes: int = 0) -> dict:
    """get_dvbs2_cfg (codings/dvb-s2/modcod_to_cfg.h): bits per symbol, slots per frame, dvbs2_code_rate_t."""
    if 1 <= modcod < 12:
        return dict(bits=2, slots=90 if shortframes else 360, rate=_QPSK[modcod - 1], constellation=0)
    if 12 <= modcod < 18:
        return dict(bits=3, slots=60 if shortframes else 240, rate=_PSK8[modcod - 12], constellation=1)
    if 18 <= modcod < 24:
        return dict(bits=4, slots=45 if shortframes else 180, rate=_APSK16[modcod - 18], constellation=2)
    raise ValueError("MODCOD not supported by the generator")


# ---- LDPC (5.3.2): systematic IRA code from the standard's address tables (satdump_amd/csrc/dvbs2_tables.inc)
def _tables():
    if "t" not in _cache:
        src = open(os.path.join(_HERE, "csrc", "dvbs2_tables.inc")).read()
        grp = [int(v) for v in re.findall(r"\d+", re.search(r"S2_GRP\[\] = \{(.*?)\};", src, re.S).group(1).split("\n", 1)[1])]
        pos_body = re.sub(r"//.*", "", re.search(r"S2_POS\[\] = \{(.*?)\};", src, re.S).group(1))
        pos = [int(v) for v in re.findall(r"\d+", pos_body)]
        tabs = []
        for m in re.finditer(r'\{"(\w+)", (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\}', src):
            name, M, N, K, cn, lt, ng, go, po, npos = m.group(1), *[int(v) for v in m.groups()[1:]]
            tabs.append(dict(name=name, M=M, N=N, K=K, groups=[(grp[2 * (go + i)], grp[2 * (go + i) + 1]) for i in range(ng)], pos=pos[po:po + npos]))
        _cache["t"] = tabs
    return _cache["t"]


def ldpc_table(framesize: int, rate_code: int) -> dict:
    normal = [0, 1, 2, 3, 4, 5, 6, 7, 8, -1, 9, 10]
    short = [11, 12, 13, 14, 15, 16, 17, 18, 19, -1, 20, -1]
    idx = (normal if framesize == 0 else short)[rate_code]
    if idx < 0:
        raise ValueError("no LDPC table for this frame size / rate")
    return _tables()[idx]


def ldpc_check_rows(framesize: int, rate_code: int):
    """(row index, data bit) of every edge between a data bit and a check row (natural row order), and the number of rows."""
    key = ("rows", framesize, rate_code)
    if key not in _cache:
        t = ldpc_table(framesize, rate_code)
        M, R = t["M"], t["N"] - t["K"]
        q = R // M
        rows, bits = [], []
        bit, p = 0, 0
        for deg, nrows in t["groups"]:
            for _ in range(nrows):
                a = np.array(t["pos"][p:p + deg], dtype=np.int64)
                p += deg
                for m in range(M):
                    rows.append((a + m * q) % R)
                    bits.append(np.full(deg, bit, dtype=np.int64))
                    bit += 1
        _cache[key] = (np.concatenate(rows), np.concatenate(bits), R)
    return _cache[key]


def ldpc_encode(framesize: int, rate_code: int, data_bits: np.ndarray) -> np.ndarray:
    """data_bits uint8 [nframes, K] -> code words uint8 [nframes, N]: parity[r] = parity[r - 1] ^ XOR(data bits of check row r)."""
    rows, bits, R = ldpc_check_rows(framesize, rate_code)
    nf = data_bits.shape[0]
    order = np.argsort(rows, kind="stable")
    srows, sbits = rows[order], bits[order]
    starts = np.searchsorted(srows, np.arange(R))
    acc = np.zeros((nf, R), dtype=np.uint8)
    for f in range(nf):
        acc[f] = np.bitwise_xor.reduceat(data_bits[f, sbits], starts)
    par = np.bitwise_xor.accumulate(acc, axis=1)
    return np.concatenate([data_bits, par], axis=1)


# ---- BCH (5.3.1): t-error-correcting, generator = product of the minimal polynomials of alpha^1, alpha^3, ... alpha^(2t - 1)
_KN = [16008, 21408, 25728, 32208, 38688, 43040, 48408, 51648, 53840, 0, 57472, 58192]
_NN = [16200, 21600, 25920, 32400, 38880, 43200, 48600, 51840, 54000, 0, 57600, 58320]
_TN = [12, 12, 12, 12, 12, 10, 12, 12, 10, 0, 8, 8]
_KS = [3072, 5232, 6312, 7032, 9552, 10632, 11712, 12432, 13152, 0, 14232, 0]
_NS = [3240, 5400, 6480, 7200, 9720, 10800, 11880, 12600, 13320, 0, 14400, 0]


def bch_dims(framesize: int, rate_code: int):
    """(kbch, nbch, t)."""
    if framesize == 0:
        return _KN[rate_code], _NN[rate_code], _TN[rate_code]
    return _KS[rate_code], _NS[rate_code], 12


def _bch_generator(m: int, prim: int, t: int) -> int:
    """GF(2)[x] polynomial (bit i = coefficient of x^i) over GF(2^m) defined by `prim`."""
    key = ("g", m, prim, t)
    if key in _cache:
        return _cache[key]
    n = (1 << m) - 1
    ex = [0] * (2 * n)
    a = 1
    for i in range(n):
        ex[i] = a
        a <<= 1
        if a >> m:
            a ^= prim
    for i in range(n, 2 * n):
        ex[i] = ex[i - n]
    lg = {ex[i]: i for i in range(n)}

    def mul(x, y):
        return 0 if x == 0 or y == 0 else ex[lg[x] + lg[y]]

    g = 1
    for k in range(1, 2 * t, 2):
        # minimal polynomial of alpha^k: product over its conjugates (x - alpha^(k 2^j))
        conj, e = [], k
        while e not in conj:
            conj.append(e)
            e = (e * 2) % n
        poly = [1]  # coefficients in GF(2^m), lowest first
        for e in conj:
            root = ex[e]
            nxt = [0] * (len(poly) + 1)
            for i, c in enumerate(poly):
                nxt[i + 1] ^= c
                nxt[i] ^= mul(c, root)
            poly = nxt
        assert all(c in (0, 1) for c in poly)
        mp = sum(1 << i for i, c in enumerate(poly) if c)
        # g *= mp over GF(2)
        r, s, mm = 0, 0, mp
        while mm:
            if mm & 1:
                r ^= g << s
            mm >>= 1
            s += 1
        g = r
    _cache[key] = g
    return g


def bch_encode(framesize: int, rate_code: int, data_bits: np.ndarray) -> np.ndarray:
    """data_bits uint8 [nframes, kbch] (first bit = highest power) -> [nframes, nbch]: data followed by the remainder of data(x) x^(n - k) by g(x)."""
    kb, nb, t = bch_dims(framesize, rate_code)
    if kb == 0:
        raise ValueError("no BCH code for this frame size / rate")
    assert data_bits.shape[1] == kb
    g = _bch_generator(16, 0x1002D, t) if framesize == 0 else _bch_generator(14, 0x402B, 12)
    npar = nb - kb
    assert g.bit_length() - 1 == npar
    key = ("R", framesize, rate_code)
    if key not in _cache:
        # R[j] = x^(npar + kb - 1 - j) mod g, as npar bits (highest power first)
        R = np.zeros((kb, npar), dtype=np.uint8)
        r = 1 << (npar - 1)  # x^(npar - 1); one more shift gives x^npar mod g = the last data bit's remainder
        top = 1 << npar
        for j in range(kb - 1, -1, -1):
            r <<= 1
            if r & top:
                r ^= g
            R[j] = [(r >> (npar - 1 - b)) & 1 for b in range(npar)]
        _cache[key] = R.astype(np.float32)
    R = _cache[key]
    par = (data_bits.astype(np.float32) @ R).astype(np.int64) & 1
    return np.concatenate([data_bits, par.astype(np.uint8)], axis=1)


# ---- BB scrambler (5.2.2): 1 + x^14 + x^15, initial state 100101010000000, the whole BBFRAME
def bb_prbs(nbytes: int) -> np.ndarray:
    key = ("prbs", nbytes)
    if key not in _cache:
        sr = 0x4A80
        out = np.zeros(nbytes, dtype=np.uint8)
        for i in range(nbytes):
            v = 0
            for _ in range(8):
                b = ((sr >> 1) ^ sr) & 1   # outputs of stages 14 and 15
                v = (v << 1) | b
                sr = (sr >> 1) | (b << 14)
            out[i] = v
        _cache[key] = out
    return _cache[key]


def bb_scramble(frames: np.ndarray) -> np.ndarray:
    """uint8 [nframes, kbch / 8] XOR the PRBS (its own inverse)."""
    return frames ^ bb_prbs(frames.shape[1])[None, :]


# ---- bit interleaver (5.3.3) and constellation mapping
def interleave(constellation: int, rate_code: int, cw: np.ndarray) -> np.ndarray:
    """code word bits [nframes, N] -> bits in symbol order [nframes, N] (what S2Deinterleaver::deinterleave, s2_deinterleaver.cpp:92-145, takes apart)."""
    nf, N = cw.shape
    if constellation == 0:
        return cw  # QPSK: none (the reference's pair swap belongs to its demapper's bit order: see map_symbols)
    bits = {1: 3, 2: 4, 3: 5}[constellation]
    rows = N // bits
    cols = cw.reshape(nf, bits, rows)  # column c = bits [c * rows, (c + 1) * rows)
    if constellation == 1 and rate_code == RATES["3/5"]:
        cols = cols[:, ::-1, :]
    return np.ascontiguousarray(cols.transpose(0, 2, 1)).reshape(nf, N)


def constellation_points(constellation: int, gamma: float = 0.0) -> np.ndarray:
    """constellation_t's point table (constellation.cpp:28-110): index = the symbol's bits, first bit = MSB, each bit INVERTED
    (tools/dvbs2_mod/main.cpp: const_bits = const_bits << 1 | !bit)."""
    if constellation == 0:
        s = np.sqrt(2.0)
        return np.array([-s - 1j * s, s - 1j * s, -s + 1j * s, s + 1j * s]) / 2.0  # unit circle
    if constellation == 1:
        r = np.sqrt(0.5)
        return np.array([-1j, -r + 1j * r, r - 1j * r, 1j, -r - 1j * r, -1.0, 1.0, r + 1j * r])
    if constellation == 2:
        g1 = gamma or 2.57
        r1 = np.sqrt(4 / (1 + 3 * g1 * g1))
        r2 = g1 * r1

        def polar(r, n, i):
            return r * np.exp(2j * np.pi * i / n)
        t = {15: (r2, 12, 1.5), 14: (r2, 12, 10.5), 13: (r2, 12, 4.5), 12: (r2, 12, 7.5), 11: (r2, 12, 0.5), 10: (r2, 12, 11.5), 9: (r2, 12, 5.5), 8: (r2, 12, 6.5),
             7: (r2, 12, 2.5), 6: (r2, 12, 9.5), 5: (r2, 12, 3.5), 4: (r2, 12, 8.5), 3: (r1, 4, 0.5), 2: (r1, 4, 3.5), 1: (r1, 4, 1.5), 0: (r1, 4, 2.5)}
        return np.array([polar(*t[k]) for k in range(16)])
    raise ValueError("constellation")


def map_symbols(constellation: int, sym_bits: np.ndarray, gamma: float = 0.0) -> np.ndarray:
    """bits in symbol order [nframes, N] -> unit-power symbols complex128 [nframes, N / bits]."""
    bits = {0: 2, 1: 3, 2: 4}[constellation]
    nf, N = sym_bits.shape
    b = (1 - sym_bits.reshape(nf, N // bits, bits)).astype(np.int64)  # inverted
    if constellation == 0:
        # the reference's QPSK receive path: demapper bit order + its "de-interleaver" pair swap put code word bit 2k on I, 2k + 1 on Q, bit 0 -> +
        return ((2.0 * b[:, :, 0] - 1.0) + 1j * (2.0 * b[:, :, 1] - 1.0)) / np.sqrt(2.0)
    v = np.zeros((nf, N // bits), dtype=np.int64)
    for i in range(bits):
        v = (v << 1) | b[:, :, i]
    return constellation_points(constellation, gamma)[v]


# ---- PL framing (5.5): header, PL scrambling
def pls_codewords() -> np.ndarray:
    """The 128 PLS code words (5.5.2.4; index = MODCOD << 2 | short << 1 | pilots): (32, 6) generator, every bit sent twice (pilots bit: the second
    copy complemented), scrambled with the standard's 64-bit sequence."""
    G = [0x55555555, 0x33333333, 0x0f0f0f0f, 0x00ff00ff, 0x0000ffff, 0xffffffff]
    out = np.zeros(128, dtype=np.uint64)
    for index in range(128):
        y = 0
        for row in range(6):
            if (index >> (6 - row)) & 1:
                y ^= G[row]
        code = 0
        for bit in range(31, -1, -1):
            yi = (y >> bit) & 1
            code = (code << 2) | (yi << 1) | ((yi ^ 1) if (index & 1) else yi)
        out[index] = code ^ 0x719d83c953422dfa
    return out


def sof_symbols() -> np.ndarray:
    """The 26 pi/2-BPSK symbols of the start-of-frame field 0x18D2E82 (5.5.2.1; dvbs2/s2_defs.h:16-36)."""
    s = np.arange(26)
    bit = (0x18d2e82 >> (25 - s)) & 1
    return np.exp(1j * (np.pi / 4 + 2 * np.pi * (bit * 2 + (s & 1)) / 4))


def pls_symbols(index: int) -> np.ndarray:
    """The 64 pi/2-BPSK symbols of PLS code word `index` (dvbs2/s2_defs.h:74-80)."""
    cw = int(pls_codewords()[index])
    i = np.arange(64)
    yi = np.array([(cw >> (63 - k)) & 1 for k in range(64)])
    nyi = yi ^ (i & 1)
    return ((1 - 2 * nyi) + 1j * (1 - 2 * yi)) / np.sqrt(2.0)


def gold_rn(count: int) -> np.ndarray:
    """PL scrambling sequence Rn in {0..3}, Gold code n = 0 (5.5.4): symbol i is multiplied by exp(j Rn pi / 2)."""
    key = ("rn",)
    if key not in _cache:
        x, y = 1, 0x3ffff
        z = np.zeros(2 * 131072, dtype=np.uint8)
        for i in range(2 * 131072):
            z[i] = (x ^ y) & 1
            x = ((((x >> 7) ^ x) & 1) << 18 | x) >> 1
            y = ((((y >> 10) ^ (y >> 7) ^ (y >> 5) ^ y) & 1) << 18 | y) >> 1
        _cache[key] = (z[:131072] | (z[131072:] << 1)).astype(np.int64)
    return _cache[key][:count]


def bbframes_random(framesize: int, rate_code: int, nframes: int, seed: int) -> np.ndarray:
    """Random BBFRAME payloads uint8 [nframes, kbch / 8] (what the receiver's .bbframe output must reproduce)."""
    kb, _, _ = bch_dims(framesize, rate_code)
    return np.random.default_rng(seed).integers(0, 256, (nframes, kb // 8), dtype=np.uint8)


def fecframes(framesize: int, rate_code: int, bb: np.ndarray) -> np.ndarray:
    """BBFRAMEs uint8 [nframes, kbch / 8] -> FECFRAME bits uint8 [nframes, 64800 / 16200] (BB scrambler, BCH, LDPC)."""
    bits = np.unpackbits(bb_scramble(bb), axis=1)
    return ldpc_encode(framesize, rate_code, bch_encode(framesize, rate_code, bits))


def plframes(modcod: int, shortframes: int, bb: np.ndarray, gamma: float = 0.0) -> np.ndarray:
    """BBFRAMEs -> PLFRAME symbols complex128 [nframes, 90 + slots * 90], unit power, no pilots."""
    c = modcod_cfg(modcod, shortframes)
    cw = fecframes(shortframes, c["rate"], bb)
    sym = map_symbols(c["constellation"], interleave(c["constellation"], c["rate"], cw), gamma)
    nsym = sym.shape[1]
    assert nsym == c["slots"] * 90
    sym = sym * np.exp(1j * np.pi / 2 * gold_rn(nsym))[None, :]
    hdr = np.concatenate([sof_symbols(), pls_symbols((modcod << 2) | (shortframes << 1))])
    return np.concatenate([np.broadcast_to(hdr, (len(bb), 90)), sym], axis=1)


def symbol_stream(frames: np.ndarray, seed: int, lead: int = 0, tail_frames: int = 2, cfo: float = 0.0, phase0: float = 0.4, esn0_db: float = 10.0, amplitude: float = 0.7) -> np.ndarray:
    """A clock-recovered symbol stream (complex64): `lead` noise symbols, the frames back to back, tail_frames frames' worth of noise (so that the last
    frame's search window is there), a rotation (cfo rad / symbol), AWGN at Es/N0, scaled to `amplitude`."""
    rng = np.random.default_rng(seed)
    raw = frames.shape[1]
    parts = [(rng.standard_normal(lead) + 1j * rng.standard_normal(lead)) * 0.7, frames.reshape(-1),
             (rng.standard_normal(tail_frames * raw) + 1j * rng.standard_normal(tail_frames * raw)) * 0.7]
    s = np.concatenate(parts)
    n = len(s)
    s = s * np.exp(1j * (cfo * np.arange(n) + phase0))
    sigma = np.sqrt(1.0 / (2.0 * 10 ** (esn0_db / 10)))
    s = (s + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))) * amplitude
    return s.astype(np.complex64)
