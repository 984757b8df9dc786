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
               nroots: int = 32, subset=None, payload=None) -> np.ndarray:
    """Random-payload CADUs: 4-byte ASM + rs_i*255 bytes (interleaved codeblock, randomised).
    subset: frame indices to build (the payload of ALL nframes is drawn, so frame i is the same whichever subset asks for it).
    payload: [n, rs_i, k] bytes to use instead of drawing them."""
    k = 255 - nroots
    if payload is not None:
        data = np.asarray(payload, dtype=np.uint8)
    else:
        rng = np.random.default_rng(seed)
        data = rng.integers(0, 256, size=(nframes, rs_i, k), dtype=np.uint8)
        if subset is not None:
            data = data[np.asarray(subset, dtype=np.int64)]
    nframes = len(data)
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


def fy3_diff_encode(dibits: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Transmit-side inverse of fengyun3::FengyunDiff::work2 (plugins/fengyun3_support/fengyun3/diff.cpp:49-78): the rail bits (x, y) = (in1, in2) whose
    decoder output is the dibit stream (hi, lo). The decoder emits (y^y', x^x') when x != y and (x^x', y^y') otherwise (x', y' = the previous pair), so from
    a = x' ^ y' ^ hi ^ lo: a == 0 -> x = x' ^ hi, y = y' ^ lo (and x == y holds), a == 1 -> x = x' ^ lo, y = y' ^ hi. Starts from (0, 0) like the decoder."""
    d = np.asarray(dibits, dtype=np.uint8).reshape(-1, 2)
    x = np.empty(len(d), dtype=np.uint8)
    y = np.empty(len(d), dtype=np.uint8)
    xp = yp = 0
    for i, (hi, lo) in enumerate(d.tolist()):
        if (xp ^ yp ^ hi ^ lo) == 0:
            xp, yp = xp ^ hi, yp ^ lo
        else:
            xp, yp = xp ^ lo, yp ^ hi
        x[i] = xp
        y[i] = yp
    return x, y


def fy3_ahrpt_soft(nframes: int, seed: int = 3, sigma: float = 18.0, amp: float = 70.0, invert_second: bool = True, branches_swapped: bool = False, lead: int = 0,
                   gaps=(), noise_tail: int = 0, mpt: bool = False):
    """A FengYun-3 AHRPT .soft stream as fengyun_ahrpt_decoder reads it (module_fengyun_ahrpt_decoder.cpp:58-126): 1024-byte CADUs (RS(255,223) x 4 dual
    basis, randomised) -> dibits -> differential encoder -> two rails, each r = 1/2 k = 7 punctured to 3/4 (c0 c1 . c1 c0 . per three bits, what
    Viterbi3_4's fymode depuncture undoes, viterbi_3_4.cpp:58-78) -> QPSK. The module exchanges I and Q, hands byte 0 of a pair to Viterbi 1 and byte 1
    (complemented when invert_second) to Viterbi 2, and takes Viterbi 2's bits as the differential decoder's first input -- unless `branches_swapped`, the
    case its noSyncRuns counter resolves. lead: garbage bytes in front, gaps: (byte position, bytes removed). Returns (soft int8, plain CADUs).
    mpt: the stream fengyun_mpt_decoder reads instead (module_fengyun_mpt_decoder.cpp:56-75): the rails at rate 1/2 unpunctured, every rail's byte pairs
    exchanged (the module exchanges them back in front of its Viterbi1_2), the second rail always complemented."""
    cadus = make_cadus(nframes, seed=seed, rs_i=4, dualbasis=True)
    x, y = fy3_diff_encode(np.unpackbits(cadus.reshape(-1)))
    v1_bits, v2_bits = (x, y) if branches_swapped else (y, x)
    rails = []
    if mpt:
        invert_second = True
    for bits in (v1_bits, v2_bits):
        if mpt:
            c = conv_encode(bits).reshape(-1, 2)[:, ::-1].reshape(-1).astype(np.float64) * 2.0 - 1.0
        else:
            c = puncture(conv_encode(bits), 2).astype(np.float64) * 2.0 - 1.0
        rails.append(c)
    n = min(len(rails[0]), len(rails[1]))
    rng = np.random.default_rng(seed + 100)
    v = np.empty(2 * n)
    v[1::2] = rails[0][:n] * amp  # byte 1 of a pair -> (after the I/Q exchange) Viterbi 1
    v[0::2] = rails[1][:n] * amp
    v = v + sigma * rng.standard_normal(len(v))
    s = np.where(v < -128.0, -127, np.where(v > 127.0, 127, np.trunc(v))).astype(np.int8)
    if invert_second:
        s[0::2] = ~s[0::2]
    for pos, cut in sorted(gaps, reverse=True):
        s = np.concatenate([s[:pos], s[pos + cut:]])
    if lead:
        s = np.concatenate([rng.integers(-60, 60, lead).astype(np.int8), s])
    if noise_tail:
        s = np.concatenate([s, rng.integers(-60, 60, noise_tail).astype(np.int8)])
    return s, make_cadus(nframes, seed=seed, rs_i=4, dualbasis=True, derand=False)


# --------------------------------------------------------------------------- modulation
# transmit masks over the r=1/2 coded stream (c0,c1 per bit) of the DVB-style punctured rates the reference's
# viterbi::puncturing::Depunc23/34/56/78 undo (src-core/common/codings/viterbi/depunc.h): rate code 1 = 2/3, 2 = 3/4, 3 = 5/6, 4 = 7/8
PUNCTURE_MASKS = {
    1: [1, 1, 0, 1],
    2: [1, 1, 0, 1, 1, 0],
    3: [1, 1, 0, 1, 1, 0, 0, 1, 1, 0],
    4: [1, 1, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1, 0],
}


M2X_BRANCHES, M2X_DELAY, M2X_STRIDE, M2X_MARKER = 36, 2048 * 36, 80, 0x27  # meteor/deint.h: INTER_BRANCH_COUNT, INTER_BRANCH_DELAY * INTER_BRANCH_COUNT, INTER_MARKER_STRIDE, INTER_MARKER


def m2x_interleave(soft: np.ndarray, marker_amp: int = 90, marker_errors: float = 0.0, seed: int = 0) -> np.ndarray:
    """What a Meteor-M2-x 80k transmitter does to the coded soft stream `soft` (the inverse of meteor::DeinterleaverReader, plugins/meteor_support/meteor/deint.cpp:
    100-133): the convolutional interleaver -- sample n of the data stream rides branch n % 36 and comes out of the receiver's de-interleaver (35 - n % 36) * 73 728
    samples later -- and an 8-sample marker (0x27, MSB first, a 1 = a NEGATIVE sample: soft_to_hard, deint.cpp:153-172) in front of every 72 data samples. The
    first 35 * 73 728 outputs of the de-interleaver are the zeros its ring starts with (or stale data): `soft` should begin with that much lead-in. Returns the
    transmitted stream, length len(soft) / 72 * 80 (whole strides only)."""
    s = np.ascontiguousarray(soft, dtype=np.int8)
    n = len(s) // 72 * 72
    idx = np.arange(n, dtype=np.int64)
    src = idx + (M2X_BRANCHES - 1 - idx % M2X_BRANCHES) * M2X_DELAY  # data[n] = out[n + (35 - n % 36) * 73728]
    # (behind the end of `soft` the branches carry noise, not zeros: a run of equal samples defeats the receiver's marker autocorrelation)
    fill = np.random.default_rng(seed + 78).integers(-60, 60, n).astype(np.int8)
    data = np.where(src < len(s), s[np.minimum(src, len(s) - 1)], fill).astype(np.int8)
    bits = np.array([(M2X_MARKER >> (7 - k)) & 1 for k in range(8)])
    mark = np.where(bits == 1, -marker_amp, marker_amp).astype(np.int8)
    out = np.empty((n // 72, M2X_STRIDE), dtype=np.int8)
    out[:, :8] = mark
    out[:, 8:] = data.reshape(-1, 72)
    if marker_errors > 0:
        rng = np.random.default_rng(seed + 77)
        flip = rng.random((n // 72, 8)) < marker_errors
        out[:, :8] = np.where(flip, -out[:, :8], out[:, :8])
    return out.reshape(-1)


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


# --------------------------------------------------------------------------- one long recording, synthesised slice by slice
def _splitmix64(z):
    """splitmix64 finaliser on int64 arrays (numpy or torch; two's-complement wrap-around, logical shifts by masking)."""
    m30, m27, m31 = (1 << 34) - 1, (1 << 37) - 1, (1 << 33) - 1
    z = (z ^ ((z >> 30) & m30)) * (-0x40A7B892E31B1A47)  # 0xBF58476D1CE4E5B9
    z = (z ^ ((z >> 27) & m27)) * (-0x6B2FB644ECCEEE15)  # 0x94D049BB133111EB
    return z ^ ((z >> 31) & m31)


class Recording:
    """ONE synthetic recording of `blocks` x `frames_per_block` CADUs whose samples are a pure function of the absolute
    sample index, so that any rank can synthesise any range [a, b) of it without the rest (bench.py shards one recording over
    the GPUs of a node; tests cut it at arbitrary places).

    Block r holds frames_per_block CADUs drawn from seed + 100 r. With NRZ-M the payload of its frame 1 is re-drawn until the
    block's bit parity is even, so the differential encoder's level is 0 at every block boundary: the level anywhere in the
    last frames of a block follows backwards from its end, without the frames in front. The convolutional encoder runs
    through (a slice is encoded from one frame early and that frame's symbols are dropped). The block sequence is periodic
    (block `blocks` = block 0), the carrier makes a whole number of turns over the recording, and the noise of sample m is a
    hash of m -- so the recording also tiles seamlessly in time, which lets a stateful engine process it step after step."""

    def __init__(self, spec: SynthSpec, frames_per_block: int, blocks: int = 1):
        self.spec = spec
        self.P = int(frames_per_block)
        self.blocks = int(blocks)
        ratio = Fraction(spec.samplerate / spec.symbolrate).limit_denominator(2000)
        self.up, self.down = ratio.numerator, ratio.denominator
        bits = 8 * (4 + 255 * spec.rs_i)
        # frames per whole puncture / symbol period: MetOp's 3/4 pattern consumes 6 mother-code bits per 3 information bits and
        # a frame is 8192 bits, so symbol boundaries and frame boundaries only coincide every 3 frames
        self.group = 1 if spec.conv == "1/2" else 3
        # (the 4/3 of the punctured code applies to the whole group: 3 frames = 24576 bits -> 32768 code bits; frame by frame it
        # does not divide)
        coded_bits = self.group * 2 * bits if spec.conv == "1/2" else self.group * bits * 4 // 3
        self.syms_per_group = coded_bits if spec.constellation == "bpsk" else coded_bits // 2
        if self.P % self.group:
            raise ValueError(f"frames_per_block must be a multiple of {self.group}")
        nsym_block = self.P // self.group * self.syms_per_group
        if (nsym_block * self.up) % self.down:
            raise ValueError("frames_per_block does not give a whole number of samples")
        self.nsym_block = nsym_block
        self.samples_per_block = nsym_block * self.up // self.down
        self.n_samples = self.samples_per_block * self.blocks
        cyc = round(spec.cfo_hz * self.n_samples / spec.samplerate)
        self.cfo = cyc * spec.samplerate / self.n_samples
        self._full = {}

    # ---- frames
    def _seed(self, b):
        return self.spec.seed + 100 * (b % self.blocks)

    def _kw(self):
        return dict(rs_i=self.spec.rs_i, dualbasis=self.spec.dualbasis, derand=self.spec.derand)

    def block_cadus(self, b):
        """All CADUs of block b as transmitted (frame 1 adjusted for even block parity when NRZ-M is on)."""
        b %= self.blocks
        if b not in self._full:
            cad = make_cadus(self.P, self._seed(b), **self._kw())
            if self.spec.nrzm and self.P >= 2:
                par = int(np.unpackbits(cad.reshape(-1)).sum()) & 1
                t = 0
                while par:
                    rng = np.random.default_rng(self._seed(b) + 7777 + t)
                    pl = rng.integers(0, 256, size=(1, self.spec.rs_i, 223), dtype=np.uint8)
                    new = make_cadus(1, 0, payload=pl, **self._kw())[0]
                    par ^= (int(np.unpackbits(cad[1]).sum()) ^ int(np.unpackbits(new).sum())) & 1
                    cad[1] = new
                    t += 1
            self._full = {b: cad}  # one block cached at a time (a block is ~134 MB at the bench sizes)
        return self._full[b]

    def plain_cadus(self, b):
        """Block b's CADUs as a decoder outputs them (de-randomised)."""
        cad = self.block_cadus(b)
        if not self.spec.derand:
            return cad.copy()
        pn = np.resize(_PN, 255 * self.spec.rs_i)
        out = cad.copy()
        out[:, 4:] ^= pn[None, :]
        return out

    def frames(self, f0, f1):
        """Transmitted CADUs of absolute frame indices [f0, f1) (periodic). Whole blocks come from block_cadus(); partial
        ones are built frame by frame -- valid for any frame but frame 1 of a block with NRZ-M, which needs the whole block."""
        out = []
        f = f0
        T = self.P * self.blocks
        while f < f1:
            b, i0 = divmod(f % T, self.P)
            i1 = min(self.P, i0 + (f1 - f))
            if (b in self._full) or (i0 == 0 and i1 == self.P) or (self.spec.nrzm and i0 <= 1 < i1):
                out.append(self.block_cadus(b)[i0:i1])
            else:
                out.append(make_cadus(self.P, self._seed(b), subset=np.arange(i0, i1), **self._kw()))
            f += i1 - i0
        return np.concatenate(out, axis=0) if out else np.zeros((0, 4 + 255 * self.spec.rs_i), dtype=np.uint8)

    def symbols(self, g0, g1):
        """Symbols of frame groups [g0, g1) (group = self.group frames); first symbol has absolute index g0 * syms_per_group."""
        G = self.group
        fr = self.frames(g0 * G - 1, g1 * G)  # one frame early: convolutional encoder state
        bits = np.unpackbits(fr.reshape(-1))
        fb = fr.shape[1] * 8
        if self.spec.nrzm:
            # level just before the first bit: 0 at every block boundary; backwards from the next boundary otherwise
            first = (g0 * G - 1) % (self.P * self.blocks)
            to_boundary = (self.P - first % self.P) % self.P  # frames from `first` to the next block boundary
            lvl = int(bits[: to_boundary * fb].sum()) & 1 if to_boundary * fb <= len(bits) else None
            if lvl is None:
                raise ValueError("slice does not reach a block boundary: cannot anchor the NRZ-M level")
            bits = ((np.cumsum(bits.astype(np.int64)) + lvl) & 1).astype(np.uint8)
        coded = conv_encode(bits)
        coded = coded[2 * fb:]  # drop the lead-in frame (its last 6 bits set the encoder state)
        if self.spec.conv == "3/4-metop":
            coded = puncture_metop(coded)
        lv = coded.astype(np.float32) * 2.0 - 1.0
        if self.spec.constellation == "bpsk":
            return lv.astype(np.complex64)
        return ((lv[0::2] + 1j * lv[1::2]) / np.float32(math.sqrt(2.0))).astype(np.complex64)

    # ---- samples
    def _bank(self):
        span = self.spec.span
        hu = rrc_impulse(float(self.up), self.spec.rrc_alpha, span) * math.sqrt(self.up)
        ntap = 2 * span
        H = np.zeros((self.up, ntap), dtype=np.float64)
        c = span * self.up
        for j in range(ntap):
            H[:, j] = hu[c + (j - span) * self.up + np.arange(self.up)]
        return H

    def synth_range(self, a: int, b: int, device=None, chunk: int = 1 << 24):
        """Samples [a, b) of the recording (0 <= a < b; b may exceed n_samples: the recording repeats). device=None: numpy
        complex64; else a torch.complex64 tensor on `device`."""
        span = self.spec.span
        spg = self.syms_per_group
        k_lo = (a * self.down) // self.up - span - 1
        k_hi = ((b - 1) * self.down) // self.up + span + 1
        g0 = k_lo // spg
        g1 = k_hi // spg + 1
        sy = self.symbols(g0, g1)
        ks = g0 * spg  # absolute index of sy[0]
        H = self._bank()
        ntap = 2 * span
        sps = self.up / self.down
        sigma = math.sqrt(sps / (2.0 * 10 ** (self.spec.esn0_db / 10)))
        w = 2 * math.pi * (self.cfo / self.spec.samplerate)
        key = (self.spec.seed * 0x9E3779B97F4A7C15 + 0x1234567) & ((1 << 63) - 1)
        off = int(round(self.spec.timing_offset * self.up))
        is_real = self.spec.constellation == "bpsk"
        if device is None:
            out = np.empty(b - a, dtype=np.complex64)
            Hf = H.astype(np.float32)
            for m0 in range(a, b, chunk):
                m1 = min(b, m0 + chunk)
                m = np.arange(m0, m1, dtype=np.int64)
                u = m * self.down + off
                p = u % self.up
                k0 = u // self.up - ks
                xr = np.zeros(m1 - m0, dtype=np.float32)
                xi = None if is_real else np.zeros(m1 - m0, dtype=np.float32)
                for j in range(ntap):
                    k = k0 - (j - span)
                    hj = Hf[p, j]
                    xr += sy.real[k] * hj
                    if xi is not None:
                        xi += sy.imag[k] * hj
                ph = m.astype(np.float64) * w + self.spec.phase0
                cs, sn = np.cos(ph).astype(np.float32), np.sin(ph).astype(np.float32)
                if xi is None:
                    yr, yi = xr * cs, xr * sn
                else:
                    yr, yi = xr * cs - xi * sn, xr * sn + xi * cs
                with np.errstate(over="ignore"):
                    z = _splitmix64((m % self.n_samples) ^ np.int64(key))
                u1 = (((z & 0xFFFFFF).astype(np.float32)) + np.float32(0.5)) * np.float32(2.0 ** -24)
                u2 = ((((z >> 24) & 0xFFFFFF).astype(np.float32)) + np.float32(0.5)) * np.float32(2.0 ** -24)
                r = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.float32(sigma)
                t = np.float32(2 * math.pi) * u2
                out[m0 - a:m1 - a] = ((yr + r * np.cos(t)) * np.float32(self.spec.amplitude)) + 1j * ((yi + r * np.sin(t)) * np.float32(self.spec.amplitude))
            return out
        import torch
        d_ar = torch.from_numpy(np.ascontiguousarray(sy.real)).to(device)
        d_ai = None if is_real else torch.from_numpy(np.ascontiguousarray(sy.imag)).to(device)
        d_H = torch.from_numpy(H.astype(np.float32)).to(device)
        out = torch.empty(b - a, dtype=torch.complex64, device=device)
        for m0 in range(a, b, chunk):
            m1 = min(b, m0 + chunk)
            m = torch.arange(m0, m1, dtype=torch.int64, device=device)
            u = m * self.down + off
            p = u % self.up
            k0 = torch.div(u, self.up, rounding_mode="floor") - ks
            xr = torch.zeros(m1 - m0, dtype=torch.float32, device=device)
            xi = None if is_real else torch.zeros(m1 - m0, dtype=torch.float32, device=device)
            for j in range(ntap):
                k = k0 - (j - span)
                hj = d_H[p, j]
                xr += d_ar[k] * hj
                if xi is not None:
                    xi += d_ai[k] * hj
            ph = m.to(torch.float64) * w + self.spec.phase0
            cs, sn = torch.cos(ph).to(torch.float32), torch.sin(ph).to(torch.float32)
            if xi is None:
                yr, yi = xr * cs, xr * sn
            else:
                yr, yi = xr * cs - xi * sn, xr * sn + xi * cs
            z = _splitmix64((m % self.n_samples) ^ key)
            u1 = ((z & 0xFFFFFF).to(torch.float32) + 0.5) * (2.0 ** -24)
            u2 = (((z >> 24) & 0xFFFFFF).to(torch.float32) + 0.5) * (2.0 ** -24)
            r = torch.sqrt(-2.0 * torch.log(u1)) * sigma
            t = (2 * math.pi) * u2
            out[m0 - a:m1 - a] = torch.complex((yr + r * torch.cos(t)) * self.spec.amplitude, (yi + r * torch.sin(t)) * self.spec.amplitude)
        return out

