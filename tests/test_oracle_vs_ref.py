"""CPU tests: pin the plain-C restatement (oracle/sd_oracle.c) against the REFERENCE's own sources
compiled in place (oracle/_ref/libsdref.so) on seeded inputs -- bit-exact, every stage.

The reference ships no golden vectors for this path, so "the reference itself run
here" is the anchor; tests/golden/ holds outputs of that build for machines without /root/reference."""
import ctypes

import numpy as np
import pytest

from oracle import pyref
from satdump_amd import synth
from tests import util


def test_tables_match_reference(ref):
    raw = ref.raw
    pn = np.array((ctypes.c_ubyte * 255).in_dll(raw, "ccsds_pn")[:])
    assert np.array_equal(pn, synth._PN)
    T = np.array((ctypes.c_ubyte * 256).in_dll(raw, "_ZN11reedsolomon11ToDualBasisE")[:])
    F = np.array((ctypes.c_ubyte * 256).in_dll(raw, "_ZN11reedsolomon13FromDualBasisE")[:])
    assert np.array_equal(T, synth._TO_DUAL) and np.array_equal(F, synth._FROM_DUAL)


def test_generator_encoders_match_reference(ref):
    c = synth.make_cadus(6, seed=11, derand=False)
    buf = np.ascontiguousarray(c[:, 4:]).copy()
    buf[:, 892:] = 0
    ref.lib.sdref_rs_encode(buf.ctypes.data_as(ctypes.c_void_p), 6, 1020, 1, 4, 0)
    assert np.array_equal(buf, c[:, 4:])
    bits = np.random.default_rng(0).integers(0, 2, 5000).astype(np.uint8)
    assert np.array_equal(ref.ccencode(bits), synth.conv_encode(bits))


@pytest.mark.parametrize("sigma", [15, 30, 45, 60, 90])
def test_concat_decoder_port_equals_ref(ref, port, sigma):
    spec, cadus, plain, syms = util.goes_case(nframes=24, seed=3)
    soft = synth.soft_from_symbols(syms, spec, sigma=sigma, seed=sigma)
    for usecheck in (0, 1):
        cfg = pyref.fec_cfg(constellation=pyref.BPSK, nrzm=1, rs_usecheck=usecheck)
        a = ref.concat_decode(cfg, soft, taps=True)
        b = port.concat_decode(cfg, soft, taps=True)
        for k in ("cadu", "vit_bits", "ber", "state", "frm_err"):
            assert np.array_equal(a[k], b[k]), k
    if sigma <= 30:
        assert util.frame_ids(a["cadu"], plain)[:5] == [0, 1, 2, 3, 4]


@pytest.mark.parametrize("const,sigma", [(pyref.QPSK, 30), (pyref.QPSK, 70), (pyref.OQPSK, 40)])
def test_concat_decoder_qpsk_port_equals_ref(ref, port, const, sigma):
    spec, cadus, plain, syms = util.npp_case(nframes=24, seed=9)
    soft = synth.soft_from_symbols(syms, spec, sigma=sigma, seed=1)
    # rotate the stream by 90 degrees so the phase search has work to do
    s2 = soft.copy()
    s2[0::2], s2[1::2] = soft[1::2], -soft[0::2]
    cfg = pyref.fec_cfg(constellation=const, nrzm=1, rs_usecheck=1)
    a = ref.concat_decode(cfg, s2, taps=True)
    b = port.concat_decode(cfg, s2, taps=True)
    for k in ("cadu", "vit_bits", "ber", "state", "frm_err"):
        assert np.array_equal(a[k], b[k]), k
    if sigma <= 30 and const == pyref.QPSK:
        assert len(a["cadu"]) >= 20


@pytest.mark.parametrize("sigma", [20, 50, 80])
def test_metop_decoder_port_equals_ref(ref, port, sigma):
    spec, cadus, plain, syms = util.metop_case(nframes=40, seed=5)
    soft = synth.soft_from_symbols(syms, spec, sigma=sigma, seed=2)
    a = ref.metop_decode(soft, taps=True)
    b = port.metop_decode(soft, taps=True)
    for k in ("cadu", "vit_bits", "ber", "state", "frm_err"):
        assert np.array_equal(a[k], b[k]), k
    if sigma == 20:
        ids = util.frame_ids(a["cadu"], plain)
        assert ids[:4] == list(range(ids[0], ids[0] + 4)) and ids[0] >= 0


def test_rs_decode_with_errors_port_equals_ref(ref, port):
    rng = np.random.default_rng(7)
    frames = synth.make_cadus(64, seed=8, derand=False)
    for f in range(64):
        nerr = f % 24  # up to 23 byte errors in one codeword: beyond t=16 exercises the failure path
        pos = rng.choice(255, size=nerr, replace=False)
        frames[f, 4 + pos * 4 + (f % 4)] ^= rng.integers(1, 256, size=nerr, dtype=np.uint8)
    for fill in (-1, 0):
        a, ea = ref.rs_decode(frames, fill_bytes=fill)
        b, eb = port.rs_decode(frames, fill_bytes=fill)
        assert np.array_equal(ea, eb) and np.array_equal(a, b)
    assert (ea == -1).any() and (ea > 0).any()


def test_deframer_port_equals_ref(ref, port):
    rng = np.random.default_rng(3)
    cadus = synth.make_cadus(30, seed=4)
    bits = np.unpackbits(cadus.reshape(-1))
    # garbage prefix, a bit slip in the middle, an inverted tail, a few bit errors in ASMs
    pre = rng.integers(0, 2, 777).astype(np.uint8)
    stream = np.concatenate([pre, bits[: 8192 * 12], bits[8192 * 12 + 3: 8192 * 20], 1 - bits[8192 * 20:]])
    stream[777 + 8192 * 5 + 7] ^= 1
    stream[777 + 8192 * 6 + 1] ^= 1
    for synced in (12, 18):
        for chunk in (4096, 12288, 1000):
            a = ref.deframer(stream, chunk=chunk, state_synced=synced)
            b = port.deframer(stream, chunk=chunk, state_synced=synced)
            assert np.array_equal(a, b) and len(a) > 20


def test_taps_port_equals_ref(ref, port):
    for fs, sr in ((6e6, 2333333), (2.7e6, 927000), (30e6, 15e6)):
        assert np.array_equal(ref.rrc_taps(fs, sr, 0.5), port.rrc_taps(fs, sr, 0.5))
    assert np.array_equal(ref.mm_bank(), port.mm_bank())
    a, i1, d1 = ref.resamp_bank(2700000, 3000000)
    b, i2, d2 = port.resamp_bank(2700000, 3000000)
    assert (i1, d1) == (i2, d2) == (9, 10) and a.shape == (9, 38) and np.array_equal(a, b)


BLOCKS = [(0, [1e-2, 1, 1, 65536]), (1, [6e6, 2333333, 0.5, 31]), (2, [0.003, 4, 1.0]), (2, [0.02, 2, 1.0]), (2, [0.003, 8, 1.0]),
          (3, [2.5714, (8.7e-3) ** 2 / 4, 0.5, 8.7e-3, 0.005]), (4, [2700000, 3000000]), (5, [0]), (6, [0]),
          (7, [2.5714, (8.7e-3) ** 2 / 4, 0.5, 8.7e-3, 0.005]), (7, [2.0, (8.7e-3) ** 2 / 4, 0.5, 8.7e-3, 0.005]),
          (8, [0.002, 3.14, -3.14]), (8, [0.05, 0.25, -0.25])]  # carrier-tracking PLL (pll_carrier_tracking.cpp, fast_trig.cpp)


@pytest.mark.parametrize("kind,params", BLOCKS)
def test_dsp_blocks_port_equals_ref(ref, port, kind, params):
    rng = np.random.default_rng(kind + 1)
    x = ((rng.standard_normal(120000) + 1j * rng.standard_normal(120000)) * 0.3).astype(np.complex64)
    if kind == 8:  # half noise alone (every octant of the arctangent, both wraps, the rate limit), half a carrier the loop locks to
        x[60000:] += (0.9 * np.exp(1j * (0.013 * np.arange(60000) + 1.0))).astype(np.complex64)
        x[7] = 0
    for chunk in (30000, 8193):
        a = ref.block(kind, params, x, chunk=chunk)
        b = port.block(kind, params, x, chunk=chunk)
        assert len(a) == len(b) and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("case", ["goes", "metop", "npp"])
def test_psk_demod_port_equals_ref(ref, port, case):
    if case == "goes":
        spec, cadus, plain, syms = util.goes_case(nframes=12)
        cfg = pyref.demod_cfg(samplerate=3e6, symbolrate=927000, constellation=pyref.BPSK, pll_bw=0.02, max_sps=3.0)
    elif case == "metop":
        spec, cadus, plain, syms = util.metop_case(nframes=16)
        cfg = pyref.demod_cfg()
    else:
        spec, cadus, plain, syms = util.npp_case(nframes=16)
        cfg = pyref.demod_cfg(samplerate=30e6, symbolrate=15e6, constellation=pyref.QPSK, pll_bw=0.002)
    x, _ = synth.modulate(syms, spec)
    a = ref.psk_demod(cfg, x)
    b = port.psk_demod(cfg, x)
    assert a["buffer_size"] == b["buffer_size"] and a["final_sps"] == b["final_sps"]
    assert np.array_equal(a["soft"], b["soft"])
    assert np.array_equal(a["syms"].view(np.uint32), b["syms"].view(np.uint32))


@pytest.mark.parametrize("samplerate,symbolrate,what", [(6e6, 927000, "decimate by 2, then 9/10"), (12e6, 1000000, "decimate by 4 alone"),
                                                        (33e6, 927000, "decimate by 8, then rational"), (70e6, 665400, "decimate by 32, then rational")])
def test_psk_demod_predecimator_port_equals_ref(ref, port, samplerate, symbolrate, what):
    """SmartResampler's power-of-two pre-decimator (smart_resampler.cpp:8-61, power_decim.cpp, decimating_fir.cpp:47-89, the plans'
    constant tap sets) in the restatement == the compiled reference's own SmartResamplerBlock, bit for bit."""
    spec = synth.SynthSpec(constellation="bpsk", samplerate=samplerate, symbolrate=symbolrate, conv="1/2", nrzm=True, esn0_db=9.0, amplitude=0.4,
                           cfo_hz=2000.0, seed=11)
    x, _ = synth.modulate(synth.frames_to_symbols(synth.make_cadus(3, seed=11), spec), spec)
    x = x[:400000]
    cfg = pyref.demod_cfg(constellation=pyref.BPSK, samplerate=samplerate, symbolrate=symbolrate, rrc_alpha=0.5, pll_bw=0.02, max_sps=3.0)
    a = ref.psk_demod(cfg, x)
    b = port.psk_demod(cfg, x)
    assert a["final_sps"] == b["final_sps"] and a["buffer_size"] == b["buffer_size"] and len(a["syms"]) > 1000, what
    assert np.array_equal(a["soft"], b["soft"]) and np.array_equal(a["syms"].view(np.uint32), b["syms"].view(np.uint32)), what


@pytest.mark.parametrize("opt", ["post_costas_dc", "dc_block"])
def test_psk_demod_dc_options_port_equals_ref(ref, port, opt):
    """The two DC-block options of psk_demod (in front of the chain: module_demod_base.cpp; behind the Costas loop:
    module_psk_demod.cpp:36-38, 127-134): restatement == compiled reference on a stream with a DC offset."""
    spec, cadus, plain, syms = util.metop_case(nframes=12)
    x, _ = synth.modulate(syms, spec)
    x = (x + np.complex64(0.05 - 0.02j)).astype(np.complex64)
    cfg = pyref.demod_cfg()
    setattr(cfg, opt, 1)
    a = ref.psk_demod(cfg, x)
    b = port.psk_demod(cfg, x)
    assert np.array_equal(a["soft"], b["soft"]) and np.array_equal(a["syms"].view(np.uint32), b["syms"].view(np.uint32))


def test_psk_demod_has_carrier_port_equals_ref(ref, port):
    """psk_demod's has_carrier chain (carrier PLL + DC block between the RRC filter and the Costas loop, Costas limit 0.2;
    module_psk_demod.cpp:93-125): restatement == compiled reference, bit for bit, and both refuse a non-BPSK constellation."""
    from tests.test_zy_demod_additions_gpu import _carrier_case
    x, kw = _carrier_case(nframes=12)
    cfg = pyref.demod_cfg(constellation=pyref.BPSK, **kw)
    a = ref.psk_demod(cfg, x)
    b = port.psk_demod(cfg, x)
    assert len(a["soft"]) > 100000 and np.array_equal(a["soft"], b["soft"])
    assert np.array_equal(a["syms"].view(np.uint32), b["syms"].view(np.uint32))
    bad = pyref.demod_cfg(constellation=pyref.QPSK, **kw)
    for orc in (ref, port):
        with pytest.raises(RuntimeError):
            orc.psk_demod(bad, x[:50000])


def test_sincos_restatement_matches_host_libm(port):
    libm = ctypes.CDLL("libm.so.6")
    for f in (libm.sinf, libm.cosf):
        f.restype = ctypes.c_float
        f.argtypes = [ctypes.c_float]
    xs = np.random.default_rng(5).uniform(-6.4, 6.4, 40000).astype(np.float32)
    bad = 0
    for v in xs:
        v = float(v)
        bad += port.raw.sdo_sinf(v) != libm.sinf(v)
        bad += port.raw.sdo_cosf(v) != libm.cosf(v)
    assert bad == 0


@pytest.mark.parametrize("name", [c[0] for c in util.SIMPLE_CASES])
@pytest.mark.parametrize("sigma,usecheck", [(15.0, 1), (28.0, 0)])
def test_simple_psk_decoder_port_equals_ref(ref, port, name, sigma, usecheck):
    """ccsds_simple_psk_decoder (module_ccsds_simple_psk_decoder.cpp:104-296): every slicer option, clean and noisy."""
    ck, soft, plain = util.simple_case(name, sigma=sigma)
    ck["constellation"] = {"bpsk": pyref.BPSK, "qpsk": pyref.QPSK}[ck["constellation"]]
    cfg = pyref.fec_cfg(decoder=2, rs_usecheck=usecheck, **ck)
    a, b = ref.simple_decode(cfg, soft), port.simple_decode(cfg, soft)
    assert a["n_deframed"] == b["n_deframed"] and np.array_equal(a["frm_err"], b["frm_err"])
    assert a["cadu"].shape == b["cadu"].shape and np.array_equal(a["cadu"], b["cadu"])
    if name in ("bpsk", "bpsk_nrzm", "qpsk_0deg", "qpsk_90deg", "qpsk_diff_swap", "qpsk_diff_noswap", "qpsk_method3") and sigma < 20:
        # the transmitted frames come back (the ASM is not RS protected, and with rs_fill_bytes = -1 the reference leaves
        # the last byte of each codeword uncorrected: allow a few bytes)
        near = [min(int(np.sum(f[4:] != p[4:])) for p in plain) for f in a["cadu"]]
        assert len(near) >= 9 and sum(d <= 4 for d in near) >= len(near) - 2


# ---- conv_rate != "1/2": Viterbi_Depunc + Depunc23/34/56/78 (SURVEY.md 8 row a13'), oracle only so far
@pytest.mark.parametrize("rate", [1, 2, 3, 4])
@pytest.mark.parametrize("const,sigma,nrzm,gap", [(pyref.QPSK, 20.0, 0, False), (pyref.BPSK, 32.0, 1, False), (pyref.OQPSK, 26.0, 0, True)])
def test_punctured_concat_decoder_port_equals_ref(ref, port, rate, const, sigma, nrzm, gap):
    sigma *= {1: 1.0, 2: 0.85, 3: 0.55, 4: 0.45}[rate]  # the weaker codes need the better channel to deliver frames at all
    soft, plain = util.punctured_case(rate, nframes=12, sigma=sigma, seed=11 + rate, nrzm=bool(nrzm), gap=gap)
    cfg = pyref.fec_cfg(constellation=const, nrzm=nrzm, rs_usecheck=1)
    a = ref.concat_decode_punc(cfg, rate, soft, taps=True)
    b = port.concat_decode_punc(cfg, rate, soft, taps=True)
    for k in ("cadu", "ber", "state", "frm_err", "vit_bits"):
        assert np.array_equal(a[k], b[k]), k
    assert a["state"].max() == 1  # the stream locks ...
    ids = [i for i in util.frame_ids(a["cadu"], plain) if i >= 0]
    if not gap:  # ... and delivers the transmitted frames (behind a noise gap the reference keeps its stale puncture phase: BER*5 stays
        assert len(ids) >= 10  # under the threshold on noise, so what it delivers there is whatever it delivers -- parity only)



@pytest.mark.parametrize("case", ["metop", "goes"])
def test_threaded_pipeline_equals_the_sequential_entries(ref, case):
    """sdref_pipeline_threaded (the reference's blocks on their own threads, the topology of pipeline_run.cpp:72-104 -- what bench.py
    and the full-size GPU tests decode whole recordings with) produces the sequential entries' soft symbols and CADUs, up to the tail
    its stop() drops at EOF."""
    from tests import util
    from satdump_amd import synth
    if case == "metop":
        spec, cadus, plain, syms = util.metop_case(nframes=120)
        d = pyref.demod_cfg(samplerate=6e6, symbolrate=2333333, constellation=pyref.QPSK, rrc_alpha=0.5, pll_bw=0.003)
        f = pyref.fec_cfg(viterbi_ber_thresold=0.28, viterbi_outsync_after=10)
    else:
        spec, cadus, plain, syms = util.goes_case(nframes=60)
        d = pyref.demod_cfg(samplerate=3e6, symbolrate=927000, constellation=pyref.BPSK, pll_bw=0.02, max_sps=3.0)
        f = pyref.fec_cfg(constellation=pyref.BPSK, nrzm=1, rs_usecheck=1)
    x, _ = synth.modulate(syms, spec)
    seq = ref.psk_demod(d, x, want_syms=False)["soft"]
    seqc = (ref.metop_decode(seq, ber_thr=0.28, outsync_after=10) if case == "metop" else ref.concat_decode(f, seq))["cadu"]
    th = ref.pipeline_threaded(d, f, 1 if case == "metop" else 0, x, keep_soft=True)
    m = min(len(seq), len(th["soft"]))
    assert m >= len(seq) - 2 * 8192 * 4 and np.array_equal(seq[:m], th["soft"][:m])
    k = min(len(seqc), len(th["cadu"]))
    assert k >= len(seqc) - 2 and k > 20 and np.array_equal(seqc[:k], th["cadu"][:k])


def test_doppler_restatement_equals_the_block():
    """oracle/sd_oracle.c's restatement of DopplerCorrectBlock::work's sample loop (sdo_doppler) against the block itself, compiled in place with libpredict
    (oracle/ref_wrap_doppler.cpp) and fed buffer by buffer like a baseband file feeds it: the block's own targets in, the same samples out, bit for bit -- at
    three points of an orbit (rising, high, setting Doppler)."""
    if not pyref.doppler_block_available():
        pytest.skip("oracle/_ref/libsdref_doppler.so not built (needs /root/reference at build time)")
    rng = np.random.default_rng(1)
    n, buf = 250_000, 30000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    seen = set()
    for st in (1704110400.0, 1704112000.0, 1704115000.0):
        y, t = pyref.doppler_block_ref(x, buf, 6e6, 1701.3e6, st)
        assert len(t) == (n + buf - 1) // buf and np.all(np.abs(t) < 0.1) and np.any(t != 0)
        y2, _ = pyref.doppler_ref(x, 0.01, buf, t)
        assert np.array_equal(y.view(np.uint32), y2.view(np.uint32))
        seen.add(float(np.round(t[0], 6)))
    assert len(seen) == 3
