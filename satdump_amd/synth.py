"""Seeded synthetic baseband generator for the CCSDS hot path.

Produces the *transmit* side of what the reference decodes, so that tests and bench.py have
inputs of the shape BASELINE.json names (there is no network for real recordings):

    random payload -> RS(255,223) I-interleaved, CCSDS dual basis -> randomiser -> [NRZ-M]
    -> k=7 r=1/2 convolutional code {79,109} -> [MetOp 3/4 puncture] -> BPSK/QPSK mapping
    -> RRC pulse shaping at a rational samples/symbol -> CFO + phase + AWGN -> cf32 / cs16

Everything is written from the published CCSDS 131.0-B definitions; conventions (bit order,
polynomial orientation, puncture pattern) are the inverse of what the reference's decoders
expect:
  * encoder register `state = (state << 1) | bit`, out_j = parity(state & poly_j)
    (reference decoder side: src-core/common/codings/viterbi/cc_encoder.cpp:92-104)
  * MetOp puncture = inverse of viterbi_3_4.cpp:84-105
  * NRZ-M decode is out = b ^ last (differential/nrzm.cpp:24-33) -> encode is a running XOR
  * soft > 0  <=> coded bit 1 (viterbi_1_2.cpp:44)

This module is NOT on the timed path and never touches oracle/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from fractions import Fraction

import numpy as np

ASM = 0x1ACFFC1D

# --------------------------------------------------------------------------- GF(256) / RS
_PRIM = 0x187  # x^8 + x^7 + x^2 + x + 1 (CCSDS)


def gf_tables():
    exp = np.zeros(512, dtype=np.int32)
    log = np.zeros(256, dtype=np.int32)
    e = 1
    for i in range(255):
        exp[i] = e
        log[e] = i
        e <<= 1
        if e & 0x100:
            e ^= _PRIM
    exp[255:510] = exp[0:255]
    return exp, log


_EXP, _LOG = gf_tables()


def gf_mul(a, b):
    a = np.asarray(a, dtype=np.int32)
    b = np.asarray(b, dtype=np.int32)
    r = _EXP[(_LOG[a] + _LOG[b]) % 255]
    return np.where((a == 0) | (b == 0), 0, r)


def _trace(x):
    t = 0
    y = x
    for _ in range(8):
        t ^= y
        y = int(gf_mul(y, y))
    return t & 1


def dual_basis_tables():
    """CCSDS dual basis: z_j = Tr(x * beta^j), beta = alpha^117, z_0 = MSB."""
    to_dual = np.zeros(256, dtype=np.uint8)
    for x in range(256):
        z = 0
        for j in range(8):
            z |= _trace(int(gf_mul(x, int(_EXP[(117 * j) % 255])))) << (7 - j)
        to_dual[x] = z
    from_dual = np.zeros(256, dtype=np.uint8)
    from_dual[to_dual] = np.arange(256, dtype=np.uint8)
    return to_dual, from_dual


_TO_DUAL, _FROM_DUAL = dual_basis_tables()


def rs_generator(nroots=32, fcr=112, gap=11):
    """g(x) = prod (x + alpha^(gap*(fcr+i))), coefficients highest order first."""
    g = np.array([1], dtype=np.int32)
    for i in range(nroots):
        root = int(_EXP[(gap * (fcr + i)) % 255])
        g = np.concatenate([g, [0]]) ^ np.concatenate([[0], gf_mul(g, root)])
    return g  # len nroots+1, g[0] == 1


def rs_encode(msgs: np.ndarray, nroots=32) -> np.ndarray:
    """Systematic RS(255, 255-nroots) over conventional-basis bytes. msgs: [n, k] uint8."""
    fcr = 112 if nroots == 32 else 120
    g = rs_generator(nroots, fcr, 11)[1:]  # drop leading 1
    n, k = msgs.shape
    par = np.zeros((n, nroots), dtype=np.int32)
    for i in range(k):
        fb = msgs[:, i].astype(np.int32) ^ par[:, 0]
        par = np.concatenate([par[:, 1:], np.zeros((n, 1), dtype=np.int32)], axis=1)
        par ^= gf_mul(fb[:, None], g[None, :])
    return np.concatenate([msgs, par.astype(np.uint8)], axis=1)


def ccsds_pn_table() -> np.ndarray:
    """255-byte CCSDS pseudo-randomiser: h(x) = x^8 + x^7 + x^5 + x^3 + 1, seed all ones."""
    reg = [1] * 8
    bits = []
    for _ in range(255 * 8):
        bits.append(reg[0])
        nb = reg[0] ^ reg[3] ^ reg[5] ^ reg[7]
        reg = reg[1:] + [nb]
    return np.packbits(np.array(bits, dtype=np.uint8))


_PN = ccsds_pn_table()


def make_cadus(nframes: int, seed: int, rs_i: int = 4, dualbasis: bool = True, derand: bool = True, asm: int = ASM,
               nroots: int = 32) -> np.ndarray:
    """Random-payload CADUs: 4-byte ASM + rs_i*255 bytes (interleaved codeblock, randomised)."""
    rng = np.random.default_rng(seed)
    k = 255 - nroots
    data = rng.integers(0, 256, size=(nframes, rs_i, k), dtype=np.uint8)
    if dualbasis:
        conv = _FROM_DUAL[data]
    else:
        conv = data
    cw = rs_encode(conv.reshape(-1, k), nroots).reshape(nframes, rs_i, 255)
    if dualbasis:
        cw = _TO_DUAL[cw]
    body = cw.transpose(0, 2, 1).reshape(nframes, 255 * rs_i)  # interleave: byte ii*I + b
    if derand:
        pn = np.resize(_PN, 255 * rs_i)
        body = body ^ pn[None, :]
    hdr = np.array([(asm >> 24) & 255, (asm >> 16) & 255, (asm >> 8) & 255, asm & 255], dtype=np.uint8)
    return np.concatenate([np.broadcast_to(hdr, (nframes, 4)), body], axis=1).astype(np.uint8)


# --------------------------------------------------------------------------- convolutional side
def nrzm_encode(bits: np.ndarray) -> np.ndarray:
    return (np.cumsum(bits.astype(np.int64)) & 1).astype(np.uint8)


def conv_encode(bits: np.ndarray, circular: bool = False) -> np.ndarray:
    """k=7 r=1/2, polys {79,109}; returns interleaved coded bits c0[0],c1[0],c0[1],..."""
    b = bits.astype(np.uint8)
    if circular:
        ext = np.concatenate([b[-6:], b])
    else:
        ext = np.concatenate([np.zeros(6, dtype=np.uint8), b])
    n = len(b)

    def tap(t):  # bit delayed by t
        return ext[6 - t:6 - t + n]

    c0 = tap(0) ^ tap(1) ^ tap(2) ^ tap(3) ^ tap(6)  # 79  = 0b1001111
    c1 = tap(0) ^ tap(2) ^ tap(3) ^ tap(5) ^ tap(6)  # 109 = 0b1101101
    out = np.empty(2 * n, dtype=np.uint8)
    out[0::2] = c0
    out[1::2] = c1
    return out


def puncture_metop(coded: np.ndarray) -> np.ndarray:
    """Mother-code bits m0..m5 per 3 info bits -> QPSK symbols A=(m0,m1), B=(m4,m3)."""
    n6 = len(coded) // 6
    m = coded[:n6 * 6].reshape(n6, 6)
    out = np.empty((n6, 4), dtype=np.uint8)
    out[:, 0] = m[:, 0]
    out[:, 1] = m[:, 1]
    out[:, 2] = m[:, 4]
    out[:, 3] = m[:, 3]
    return out.reshape(-1)


# --------------------------------------------------------------------------- modulation
# transmit masks over the r=1/2 coded stream (c0,c1 per bit) of the DVB-style punctured rates the reference's
# viterbi::puncturing::Depunc23/34/56/78 undo (src-core/common/codings/viterbi/depunc.h): rate code 1 = 2/3, 2 = 3/4, 3 = 5/6, 4 = 7/8
PUNCTURE_MASKS = {
    1: [1, 1, 0, 1],
    2: [1, 1, 0, 1, 1, 0],
    3: [1, 1, 0, 1, 1, 0, 0, 1, 1, 0],
    4: [1, 1, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1, 0],
}


def puncture(coded: np.ndarray, rate: int) -> np.ndarray:
    """Drop the punctured positions of an r=1/2 coded bit stream (rate code as in PUNCTURE_MASKS)."""
    m = np.asarray(PUNCTURE_MASKS[rate], dtype=bool)
    n = len(coded) // len(m) * len(m)
    return coded[:n].reshape(-1, len(m))[:, m].reshape(-1)


def rrc_impulse(sps: float, alpha: float, span: int) -> np.ndarray:
    """Root-raised-cosine impulse response sampled at `sps` samples/symbol, unit energy."""
    n = int(round(span * sps))
    t = (np.arange(-n, n + 1, dtype=np.float64)) / sps
    h = np.zeros_like(t)
    for i, ti in enumerate(t):
        if abs(ti) < 1e-12:
            h[i] = 1.0 - alpha + 4 * alpha / math.pi
        elif abs(abs(4 * alpha * ti) - 1.0) < 1e-9:
            h[i] = (alpha / math.sqrt(2)) * ((1 + 2 / math.pi) * math.sin(math.pi / (4 * alpha)) + (1 - 2 / math.pi) * math.cos(math.pi / (4 * alpha)))
        else:
            h[i] = (math.sin(math.pi * ti * (1 - alpha)) + 4 * alpha * ti * math.cos(math.pi * ti * (1 + alpha))) / (math.pi * ti * (1 - (4 * alpha * ti) ** 2))
    return h / math.sqrt(np.sum(h ** 2))


@dataclass
class SynthSpec:
    constellation: str = "bpsk"      # bpsk | qpsk
    samplerate: float = 3e6
    symbolrate: float = 927e3
    rrc_alpha: float = 0.5
    conv: str = "1/2"                # "1/2" | "3/4-metop"
    nrzm: bool = True
    rs_i: int = 4
    derand: bool = True
    dualbasis: bool = True
    amplitude: float = 0.5
    cfo_hz: float = 1000.0
    phase0: float = 0.3
    esn0_db: float = 7.0
    seed: int = 2
    span: int = 12                   # pulse half-length in symbols
    timing_offset: float = 0.0       # fractional-symbol timing offset of the first symbol


def frames_to_symbols(cadus: np.ndarray, spec: SynthSpec, circular: bool = False) -> np.ndarray:
    bits = np.unpackbits(cadus.reshape(-1))
    if spec.nrzm:
        bits = nrzm_encode(bits)
    coded = conv_encode(bits, circular=circular)
    if spec.conv == "3/4-metop":
        coded = puncture_metop(coded)
    lv = coded.astype(np.float64) * 2.0 - 1.0
    if spec.constellation == "bpsk":
        return lv.astype(np.complex128)
    lv = lv[: (len(lv) // 2) * 2]
    return (lv[0::2] + 1j * lv[1::2]) / math.sqrt(2.0)


def modulate(symbols: np.ndarray, spec: SynthSpec, periodic: bool = False, noise: bool = True, xp=np):
    """Pulse-shape `symbols` at samplerate/symbolrate (a rational number of samples/symbol) and apply
    the channel (amplitude, CFO, phase, AWGN). Returns (complex64 samples, cfo actually used).

    y[m] = sum_k a[k] h(m*down/up - k) with h a unit-energy RRC (symbol period 1), evaluated as a
    polyphase gather: phase p = (m*down) % up, k0 = (m*down) // up. With periodic=True the symbol
    sequence wraps around and the CFO is snapped to an integer number of cycles per block, so the block
    can be tiled seamlessly. Average signal power per sample is amplitude^2; Es/N0 is defined at the
    output of a matched filter."""
    ratio = Fraction(spec.samplerate / spec.symbolrate).limit_denominator(2000)
    up, down = ratio.numerator, ratio.denominator
    span = spec.span
    hu = rrc_impulse(float(up), spec.rrc_alpha, span) * math.sqrt(up)  # h(t) sampled at t = i/up, int h^2 dt = 1
    # H[p, j] = h(j + p/up) for j in [-span, span)
    ntap = 2 * span
    H = np.zeros((up, ntap), dtype=np.float64)
    c = span * up
    for j in range(ntap):
        idx = c + (j - span) * up + np.arange(up)
        H[:, j] = hu[idx]
    nsym = len(symbols)
    nout = (nsym * up) // down
    m = np.arange(nout, dtype=np.int64)
    off = int(round(spec.timing_offset * up))
    u = m * down + off
    p = u % up
    k0 = u // up
    x = np.zeros(nout, dtype=np.complex128)
    a = np.asarray(symbols)
    for j in range(ntap):
        k = k0 - (j - span)
        if periodic:
            ak = a[k % nsym]
        else:
            valid = (k >= 0) & (k < nsym)
            ak = np.where(valid, a[np.clip(k, 0, nsym - 1)], 0)
        x += ak * H[p, j]
    cfo = spec.cfo_hz
    if periodic:
        cyc = round(cfo * nout / spec.samplerate)
        cfo = cyc * spec.samplerate / nout
    rot = np.exp(1j * (2 * math.pi * (cfo / spec.samplerate) * m.astype(np.float64) + spec.phase0))
    x = x * rot
    if noise:
        rng = np.random.default_rng(spec.seed + 7919)
        sps = up / down
        sigma = math.sqrt(sps / (2.0 * 10 ** (spec.esn0_db / 10)))
        x = x + sigma * (rng.standard_normal(nout) + 1j * rng.standard_normal(nout))
    x = x * spec.amplitude
    return x.astype(np.complex64), cfo


def to_cs16(x: np.ndarray) -> np.ndarray:
    out = np.empty(2 * len(x), dtype=np.int16)
    out[0::2] = np.clip(np.rint(x.real * 32767.0), -32767, 32767).astype(np.int16)
    out[1::2] = np.clip(np.rint(x.imag * 32767.0), -32767, 32767).astype(np.int16)
    return out


def soft_from_symbols(symbols: np.ndarray, spec: SynthSpec, sigma: float, seed: int, scale: float | None = None) -> np.ndarray:
    """Directly synthesise a .soft stream (int8) from symbols for FEC-only tests:
    BPSK 1 B/symbol (x50), QPSK 2 B/symbol (x100, unit-power symbols)."""
    rng = np.random.default_rng(seed)
    if spec.constellation == "bpsk":
        sc = 50.0 if scale is None else scale
        v = symbols.real * sc + sigma * rng.standard_normal(len(symbols))
    else:
        sc = 100.0 if scale is None else scale
        v = np.empty(2 * len(symbols))
        v[0::2] = symbols.real * sc
        v[1::2] = symbols.imag * sc
        v = v + sigma * rng.standard_normal(len(v))
    # module_demod_base.h:106-113 clamp semantics
    out = np.where(v < -128.0, -127, np.where(v > 127.0, 127, np.trunc(v)))
    return out.astype(np.int8)


# --------------------------------------------------------------------------- device-side modulator (bench inputs)
def modulate_torch(symbols: np.ndarray, spec: SynthSpec, device, periodic: bool = True, noise_seed: int | None = None,
                   chunk: int = 1 << 24):
    """Same signal model as modulate(), evaluated with torch on `device` so that multi-GB inputs of BASELINE.json's
    configs can be synthesised directly in HBM (float32 pulse shaping, float64 carrier phase). Returns
    (torch.complex64 tensor [nout], cfo used). NOT on any timed path."""
    import torch

    ratio = Fraction(spec.samplerate / spec.symbolrate).limit_denominator(2000)
    up, down = ratio.numerator, ratio.denominator
    span = spec.span
    hu = rrc_impulse(float(up), spec.rrc_alpha, span) * math.sqrt(up)
    ntap = 2 * span
    H = np.zeros((up, ntap), dtype=np.float64)
    c = span * up
    for j in range(ntap):
        H[:, j] = hu[c + (j - span) * up + np.arange(up)]
    nsym = len(symbols)
    nout = (nsym * up) // down
    is_real = np.isrealobj(symbols) or not np.any(np.imag(symbols))
    a = np.asarray(symbols)
    d_ar = torch.from_numpy(np.ascontiguousarray(a.real, dtype=np.float32)).to(device)
    d_ai = None if is_real else torch.from_numpy(np.ascontiguousarray(a.imag, dtype=np.float32)).to(device)
    d_H = torch.from_numpy(H.astype(np.float32)).to(device)
    cfo = spec.cfo_hz
    if periodic:
        cyc = round(cfo * nout / spec.samplerate)
        cfo = cyc * spec.samplerate / nout
    w = 2 * math.pi * (cfo / spec.samplerate)
    sps = up / down
    sigma = math.sqrt(sps / (2.0 * 10 ** (spec.esn0_db / 10)))
    gen = torch.Generator(device=device)
    gen.manual_seed((spec.seed if noise_seed is None else noise_seed) + 7919)
    out = torch.empty(nout, dtype=torch.complex64, device=device)
    off = int(round(spec.timing_offset * up))
    for m0 in range(0, nout, chunk):
        m1 = min(nout, m0 + chunk)
        m = torch.arange(m0, m1, dtype=torch.int64, device=device)
        u = m * down + off
        p = u % up
        k0 = torch.div(u, up, rounding_mode="floor")
        xr = torch.zeros(m1 - m0, dtype=torch.float32, device=device)
        xi = None if is_real else torch.zeros(m1 - m0, dtype=torch.float32, device=device)
        for j in range(ntap):
            k = k0 - (j - span)
            if periodic:
                k = k % nsym
                hj = d_H[p, j]
            else:
                valid = (k >= 0) & (k < nsym)
                k = k.clamp(0, nsym - 1)
                hj = d_H[p, j] * valid
            xr += d_ar[k] * hj
            if xi is not None:
                xi += d_ai[k] * hj
        ph = (m.to(torch.float64) * w + spec.phase0)
        cs, sn = torch.cos(ph).to(torch.float32), torch.sin(ph).to(torch.float32)
        if xi is None:
            yr, yi = xr * cs, xr * sn
        else:
            yr, yi = xr * cs - xi * sn, xr * sn + xi * cs
        yr += sigma * torch.randn(m1 - m0, dtype=torch.float32, device=device, generator=gen)
        yi += sigma * torch.randn(m1 - m0, dtype=torch.float32, device=device, generator=gen)
        out[m0:m1] = torch.complex(yr * spec.amplitude, yi * spec.amplitude)
    return out, cfo
