"""meteor_lrpt_decoder on the device (the Viterbi27-based plugin decoders; satdump_amd/csrc/lrpt_decoder.hip) against the reference module's
own loop on the reference's own classes (oracle/ref_wrap.cpp: sdref_lrpt_decode -- Correlator, rotate_soft, Viterbi27, NRZMDiff, derand_ccsds, ReedSolomon
compiled in place): byte work, bit-exact CADUs -- clean and noisy streams, every constellation turn and I/Q swap the correlator resolves, NRZ-M, garbage in
front, bytes missing in the middle (the correlator slides the frame), pure noise, any cut of the stream into calls."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyref
from satdump_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import torch
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")
    from satdump_amd import capi as c
    c.lib()
    return c


def lrpt_soft(nframes, seed=3, diff=False, sigma=18.0, turn=0, swap=False, lead=0, gaps=(), noise_tail=0):
    """An LRPT .soft stream: 1024-byte CADUs (RS(255,223) x 4 in the conventional basis, randomised), NRZ-M if `diff`, r = 1/2 k = 7, QPSK with I = c0, Q = c1,
    x100 + noise. turn: quarter turns of the constellation, swap: I and Q exchanged, lead: garbage bytes in front, gaps: (byte position, bytes removed)."""
    cadus = synth.make_cadus(nframes, seed=seed, rs_i=4, dualbasis=False)
    bits = np.unpackbits(cadus.reshape(-1))
    if diff:
        bits = synth.nrzm_encode(bits)
    coded = synth.conv_encode(bits).astype(np.float64) * 2.0 - 1.0
    i, q = coded[0::2].copy(), coded[1::2].copy()
    for _ in range(turn % 4):
        i, q = -q, i
    if swap:
        i, q = q, i
    rng = np.random.default_rng(seed + 100)
    v = np.empty(2 * len(i))
    v[0::2], v[1::2] = i * 70.0, q * 70.0
    v = v + sigma * rng.standard_normal(len(v))
    s = np.where(v < -128.0, -127, np.where(v > 127.0, 127, np.trunc(v))).astype(np.int8)
    for pos, cut in sorted(gaps, reverse=True):
        s = np.concatenate([s[:pos], s[pos + cut:]])
    if lead:
        s = np.concatenate([rng.integers(-60, 60, lead).astype(np.int8), s])
    if noise_tail:
        s = np.concatenate([s, rng.integers(-60, 60, noise_tail).astype(np.int8)])
    plain = synth.make_cadus(nframes, seed=seed, rs_i=4, dualbasis=False, derand=False)
    return s, plain


def run_engine(capi, to_dev, to_host, zeros_dev, soft, diff, cuts=None):
    cfg = capi.LrptCfg()
    capi.lib().sdhip_lrpt_cfg_default(C.byref(cfg))
    cfg.diff_decode = int(diff)
    h = capi.lib().sdhip_lrpt_create(C.byref(cfg))
    assert h, capi.last_error()
    out = []
    cuts = cuts or [0, len(soft)]
    seen_lock = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        d_in = to_dev(soft[a:b]) if b > a else zeros_dev(16, np.int8)
        cap = (b - a) // 16384 + 4
        d_out = zeros_dev(cap * 1024, np.uint8)
        n = capi.lib().sdhip_lrpt_process_dev(h, C.c_void_p(d_in[1]), b - a, C.c_void_p(d_out[1]), cap)
        assert n >= 0, capi.last_error()
        out.append(to_host(d_out)[: n * 1024].reshape(n, 1024).copy())
        st = capi.LrptStats()
        capi.lib().sdhip_lrpt_get_stats(h, C.byref(st))
        seen_lock.append(st.correlator_lock)
    st = capi.LrptStats()
    capi.lib().sdhip_lrpt_get_stats(h, C.byref(st))
    capi.lib().sdhip_lrpt_destroy(h)
    return np.concatenate(out, axis=0), st


CASES = [
    dict(nframes=12),
    dict(nframes=10, turn=1), dict(nframes=10, turn=2), dict(nframes=10, turn=3),
    dict(nframes=10, swap=True), dict(nframes=10, swap=True, turn=1), dict(nframes=10, swap=True, turn=3),
    dict(nframes=14, diff=True, sigma=25.0),
    dict(nframes=12, lead=3334, sigma=30.0),
    dict(nframes=16, lead=10, gaps=((5 * 16384 + 7000, 1236), (11 * 16384, 16000))),
    dict(nframes=8, sigma=60.0),     # at the edge: RS decides frame by frame
    dict(nframes=3, noise_tail=6 * 16384 + 500),
    dict(nframes=4, noise_tail=14 * 16384 + 777, lead=5000),  # more slides than speculation rounds: the serial chain takes over behind the frames found so far
]


def check_decoder(capi, to_dev, to_host, zeros_dev, case):
    kw = dict(case)
    diff = kw.get("diff", False)
    soft, plain = lrpt_soft(**kw)
    want = pyref.ref().lrpt_decode(soft, diff)["cadu"]
    # one call, and the stream cut raggedly (a frame and its slide straddling calls)
    n = len(soft)
    for cuts in ([0, n], [0, 5, 16384, 16385, 40000, 40000, n // 2 + 11, n]):
        got, st = run_engine(capi, to_dev, to_host, zeros_dev, soft, diff, cuts)
        # the module's extra iterations on a stale buffer at the end of a file are not reproduced (lrpt_decoder.hip): the reference may write the last CADU
        # once more, or decode one more frame out of the part-stale last buffer
        assert len(got) <= len(want) <= len(got) + 2, (len(got), len(want))
        assert np.array_equal(got, want[: len(got)])
        assert st.frames_out == len(got) and st.soft_in == n
    if kw.get("sigma", 18.0) < 40 and not kw.get("noise_tail"):
        # every transmitted frame that lies whole in the stream comes out, in order, derandomised
        sent = [p[4:].tobytes() for p in plain]  # (the module writes its own marker, 1D CF FC 1D: module_meteor_lrpt_decoder.cpp:255-257)
        # (a mirrored constellation comes out of the reference's correlator + decoder as the COMPLEMENT of the frames -- the complement of an RS code word is
        # a code word --, and the module's own test for that looks at one byte only: same here, what counts is equality with the reference above)
        key = lambda g: g[4:].tobytes() if g[4:].tobytes() in sent else (~g[4:]).tobytes()
        ids = [sent.index(key(g)) for g in got if key(g) in sent]
        assert len(ids) == len(got) and ids == sorted(ids) and len(got) >= kw["nframes"] - 4 - 2 * len(kw.get("gaps", ()))
    return got, want


def _torch_helpers():
    import torch

    def to_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return (t, t.data_ptr())

    def zeros_dev(n, dt):
        t = torch.zeros(n, dtype={np.int8: torch.int8, np.uint8: torch.uint8}[dt], device="cuda")
        return (t, t.data_ptr())

    return to_dev, (lambda d: d[0].cpu().numpy()), zeros_dev


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_lrpt_decoder(capi, case):
    if not hasattr(pyref.ref().lib, "sdref_lrpt_decode"):
        pytest.skip("needs the compiled reference")
    check_decoder(capi, *_torch_helpers(), case)


def test_lrpt_host_path(capi):
    """push / pull with host buffers == the device path"""
    soft, _ = lrpt_soft(9, lead=222)
    want = pyref.ref().lrpt_decode(soft, False)["cadu"]
    cfg = capi.LrptCfg()
    capi.lib().sdhip_lrpt_cfg_default(C.byref(cfg))
    h = capi.lib().sdhip_lrpt_create(C.byref(cfg))
    got = []
    for a in range(0, len(soft), 50001):
        blk = np.ascontiguousarray(soft[a:a + 50001])
        assert capi.lib().sdhip_lrpt_push(h, blk.ctypes.data_as(C.c_void_p), len(blk)) == 0, capi.last_error()
        buf = np.zeros((8, 1024), dtype=np.uint8)
        k = capi.lib().sdhip_lrpt_pull(h, buf.ctypes.data_as(C.c_void_p), 8)
        got.append(buf[:k].copy())
    capi.lib().sdhip_lrpt_destroy(h)
    got = np.concatenate(got)
    assert len(got) >= 8 and np.array_equal(got, want[: len(got)])


# ---------------------------------------------------------------------------------------------------- m2x_mode + interleaved (round 6)
def m2x_run_hip(capi, tx, cuts=None, diff=True, thr=0.3, outsync=20):
    """the interleaved .soft stream through the concatenated decoder's handle with m2x_interleaved (host entries: push / flush / pull)"""
    dec = capi.FecDecoder(capi.fec_cfg(decoder=capi.DEC_CONV_CONCAT, constellation="oqpsk", cadu_size=8192, viterbi_outsync_after=outsync, viterbi_ber_thresold=thr,
                                       nrzm=1 if diff else 0, derandomize=1, derand_after_rs=0, derand_start=4, rs_i=4, rs_fill_bytes=-1, rs_dualbasis=0, rs_type=capi.RS223,
                                       rs_usecheck=1, m2x_interleaved=1))
    cuts = cuts or [0, len(tx)]
    taps = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        dec.push(tx[a:b])
        taps.append(dec.block_taps())
    dec.flush()
    taps.append(dec.block_taps())
    out = dec.pull()
    st = dec.stats()
    dec.close()
    return out, np.concatenate([t[0] for t in taps]), np.concatenate([t[1] for t in taps]), st


M2X_CASES = [
    ("clean", dict(nframes=40), None),
    ("ragged_calls", dict(nframes=40), "cuts"),
    ("marker_errors", dict(nframes=40, marker_errors=0.08), None),          # markers with bit errors: the autocorrelation still finds them
    ("misaligned_start", dict(nframes=40, lead_cut=37), None),              # the recording starts 37 samples into a marker period: the first read re-aligns (offset != 0)
    ("slip", dict(nframes=60, slip=(3_100_003, 5)), None),                  # five samples missing mid-stream: a read finds its marker early, the de-interleaver steps back
    ("quarter_turn", dict(nframes=40, turn=1), None),                       # the constellation a quarter turn on: the SECOND reader / Viterbi pair is the one that locks
]


@pytest.mark.parametrize("name,kw,mode", M2X_CASES)
def test_lrpt_m2x_interleaved(capi, name, kw, mode):
    """meteor_lrpt_decoder, m2x_mode + interleaved (module_meteor_lrpt_decoder.cpp:103-199, deint.cpp) on the device -- two de-interleaver readers as gathers, two Viterbis,
    the locked one's bits to the deframer -- against the module's loop on the reference's own classes WITH ITS SAMPLE READER PUT RIGHT (oracle/ref_wrap_lrpt_m2x.cpp,
    reader_returns = 1; as it stands in the reference tree the branch decodes nothing: tests/test_lrpt_m2x_reference_cpu.py): CADUs byte for byte, and per iteration the
    state and BER figure of the Viterbi the module takes."""
    from tests.test_lrpt_m2x_reference_cpu import LEAD
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_lrpt_m2x_decode")):
        pytest.skip("oracle/_ref/libsdref.so (with the m2x wrapper) not built")
    kw = dict(kw)
    lead_cut, slip, turn = kw.pop("lead_cut", 0), kw.pop("slip", None), kw.pop("turn", 0)
    me = kw.pop("marker_errors", 0.0)
    soft, plain = lrpt_soft(kw["nframes"], seed=7, diff=True, sigma=18.0)
    pre = np.random.default_rng(8).integers(-60, 60, LEAD).astype(np.int8)
    tx = synth.m2x_interleave(np.concatenate([pre, soft]), marker_amp=-90, marker_errors=me, seed=7)
    if turn:  # a receiver locked a quarter turn off: every (I, Q) pair of the stream turned
        t = tx[: len(tx) // 2 * 2].reshape(-1, 2).astype(np.int16)
        tx = np.stack([-t[:, 1], t[:, 0]], axis=1).reshape(-1).clip(-127, 127).astype(np.int8)
    if lead_cut:
        tx = tx[lead_cut:]
    if slip:
        tx = np.concatenate([tx[: slip[0]], tx[slip[0] + slip[1]:]])
    want = pyref.ref().lrpt_m2x_decode(tx, diff_decode=True, interleaved=True, reader_returns=1)
    cuts = [0, 5, 8192, 100_000, 100_001, len(tx) // 2, len(tx)] if mode == "cuts" else None
    got, ber, state, st = m2x_run_hip(capi, tx, cuts=cuts)
    assert len(want["cadu"]) >= kw["nframes"] - 12, len(want["cadu"])
    assert got.shape == want["cadu"].shape and np.array_equal(got, want["cadu"]), (name, got.shape, want["cadu"].shape)
    assert len(state) == want["iterations"] and np.array_equal(state, want["state"]), (name, len(state), want["iterations"])
    assert np.array_equal(ber.view(np.uint32), want["ber"].view(np.uint32)), name
    if name == "quarter_turn":
        assert (want["which"] == 2).any()  # the reference took its second Viterbi at least once (both lock: each reader turns its stream by what its markers say)
