"""GPU parity tests of the FEC chain (run with -m gpu on an MI355X): every call goes through the C ABI of
libsdhip.so and is compared, bit for bit, with the oracle (the compiled reference when oracle/_ref is present,
else the restatement)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyref
from satdump_amd import synth
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (no CPU fallback exists)"
    return torch


@pytest.fixture(scope="module")
def capi():
    from satdump_amd import capi as c
    c.lib()
    return c


@pytest.fixture(scope="module")
def orc():
    return pyref.best()


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _symbols(rng, F, nb, kind):
    stride = 2 * (F + 6)
    bits = rng.integers(0, 2, nb * F + 64).astype(np.uint8)
    coded = synth.conv_encode(bits)[: nb * 2 * F]
    if kind == "noise":
        soft = rng.integers(-127, 128, nb * 2 * F)
    elif kind == "saturated":  # full-scale symbols: exercises the uint8 metric wrap of the generic ACS kernel
        soft = (coded.astype(np.int64) * 2 - 1) * 127
        flip = rng.random(len(soft)) < 0.04
        soft = np.where(flip, -soft, soft)
    else:
        amp, sig = {"clean": (60, 20), "noisy": (50, 45)}[kind]
        soft = np.clip(np.rint((coded.astype(float) * 2 - 1) * amp + rng.standard_normal(len(coded)) * sig), -127, 127).astype(np.int64)
    u = soft + 127
    u[u == 128] = 127
    if kind == "noisy":
        u[rng.random(len(u)) < 0.1] = 128  # sprinkle erasures
    syms = np.full(nb * stride, 128, dtype=np.uint8)
    for b in range(nb):
        syms[b * stride: b * stride + 2 * F] = u[b * 2 * F:(b + 1) * 2 * F]
    return syms


@pytest.mark.parametrize("F,nb", [(4096, 9), (12288, 4), (1024, 5), (5116, 3), (640, 7)])
@pytest.mark.parametrize("kind", ["clean", "noisy", "noise", "saturated"])
def test_ccdecoder_bit_exact(torch_cuda, capi, orc, F, nb, kind):
    """k_vit_decode == viterbi::CCDecoder::work chained over blocks (cc_decoder.cpp:295-302)."""
    rng = np.random.default_rng(F + nb)
    syms = _symbols(rng, F, nb, kind)
    want = orc.ccdecoder(F, syms)
    d_syms = _dev(torch_cuda, syms)
    d_out = torch_cuda.zeros(nb * F, dtype=torch_cuda.uint8, device="cuda")
    rc = capi.lib().sdhip_op_ccdecoder(0, F, C.c_void_p(d_syms.data_ptr()), nb, C.c_void_p(d_out.data_ptr()))
    assert rc == 0, capi.last_error()
    got = d_out.cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("seg", [1024, 2048])
@pytest.mark.parametrize("kind", ["noisy", "noise"])
def test_ccdecoder_long_segments(torch_cuda, capi, orc, seg, kind, monkeypatch):
    """The lane-per-segment decoder with 1024 / 2048 trellis steps per lane (what large batches pick: launch_vit_decode2), forced here on
    a small batch through SDHIP_VIT2_SEG: bit-identical to CCDecoder::work like the 512-step default."""
    monkeypatch.setenv("SDHIP_VIT2_SEG", str(seg))
    F, nb = 12288, 5
    rng = np.random.default_rng(seg)
    syms = _symbols(rng, F, nb, kind)
    want = orc.ccdecoder(F, syms)
    d_syms = _dev(torch_cuda, syms)
    d_out = torch_cuda.zeros(nb * F, dtype=torch_cuda.uint8, device="cuda")
    rc = capi.lib().sdhip_op_ccdecoder(0, F, C.c_void_p(d_syms.data_ptr()), nb, C.c_void_p(d_out.data_ptr()))
    assert rc == 0, capi.last_error()
    assert np.array_equal(d_out.cpu().numpy(), want)


@pytest.mark.parametrize("fill", [-1, 0])
def test_rs_decode_bit_exact(torch_cuda, capi, orc, fill):
    """k_rs == ReedSolomon::decode_interlaved incl. the >t failure / miscorrection boundary (decode.c:32-145)."""
    rng = np.random.default_rng(7)
    nfr = 96
    frames = synth.make_cadus(nfr, seed=8, derand=False)
    for f in range(nfr):
        nerr = f % 24
        pos = rng.choice(255, size=nerr, replace=False)
        frames[f, 4 + pos * 4 + (f % 4)] ^= rng.integers(1, 256, size=nerr, dtype=np.uint8)
    want, ewant = orc.rs_decode(frames, fill_bytes=fill)
    d = _dev(torch_cuda, frames)
    d_err = torch_cuda.zeros(nfr * 4, dtype=torch_cuda.int32, device="cuda")
    rc = capi.lib().sdhip_op_rs_decode(0, C.c_void_p(d.data_ptr() + 4), nfr, 1024, 1, 4, capi.RS223, fill, C.c_void_p(d_err.data_ptr()))
    assert rc == 0, capi.last_error()
    assert np.array_equal(d_err.cpu().numpy().reshape(nfr, 4), ewant)
    assert np.array_equal(d.cpu().numpy(), want)
    assert (ewant == -1).any() and (ewant > 8).any()


def _run_dev(torch, capi, cfg, soft, chunks=None):
    dec = capi.FecDecoder(cfg)
    d_soft = _dev(torch, soft)
    cap = len(soft) // 4096 + 16
    outs, bers, states = [], [], []
    bounds = [0, len(soft)] if not chunks else chunks
    for a, b in zip(bounds[:-1], bounds[1:]):
        d_out = torch.zeros((cap, dec.cadu_bytes), dtype=torch.uint8, device="cuda")
        n = dec.process_dev(d_soft.data_ptr() + a, b - a, d_out.data_ptr(), cap)
        outs.append(d_out[:n].cpu().numpy())
        be, st = dec.block_taps()
        bers.append(be)
        states.append(st)
    return np.concatenate(outs), np.concatenate(bers), np.concatenate(states), dec.stats()


@pytest.mark.parametrize("sigma", [15, 30, 45, 60, 90])
@pytest.mark.parametrize("usecheck", [0, 1])
def test_concat_decoder_bpsk(torch_cuda, capi, orc, sigma, usecheck):
    """GOES-HRIT-like (BASELINE config 2 in miniature): BPSK r=1/2 + NRZ-M + RS(255,223) I=4, from clean to unlocked."""
    spec, cadus, plain, syms = util.goes_case(nframes=40, seed=3)
    soft = synth.soft_from_symbols(syms, spec, sigma=sigma, seed=sigma)
    oc = pyref.fec_cfg(constellation=pyref.BPSK, nrzm=1, rs_usecheck=usecheck)
    want = orc.concat_decode(oc, soft)
    cfg = capi.fec_cfg(constellation="bpsk", nrzm=1, rs_i=4, rs_type=capi.RS223, rs_usecheck=usecheck)
    got, ber, state, st = _run_dev(torch_cuda, capi, cfg, soft)
    assert np.array_equal(state, want["state"])
    assert np.array_equal(ber.view(np.uint32), want["ber"].view(np.uint32))
    assert got.shape == want["cadu"].shape and np.array_equal(got, want["cadu"])
    if sigma <= 30:
        assert util.frame_ids(got, plain)[:6] == [0, 1, 2, 3, 4, 5]


@pytest.mark.parametrize("rs_i,usecheck", [(1, 0), (1, 1), (2, 0)])
def test_short_cadus_and_the_fill_bytes_overrun(torch_cuda, capi, orc, rs_i, usecheck):
    """CADUs shorter than half a Viterbi buffer (the 2048 / 2072-bit pipelines): the reference's deframer returns two frames from
    one call, and with rs_fill_bytes = -1 (the module's default) ReedSolomon's interleave loop runs one byte past every codeblock
    (reedsolomon.cpp:145-156): the first rs_i bytes of the NEXT frame come out as the first corrected message byte of each
    codeword. Found by tools/twin/fec_fuzz2.py (seed 8); the output must match byte for byte, sync-marker bytes included."""
    rng = np.random.default_rng(5 + rs_i)
    cadus = synth.make_cadus(24, seed=77 + rs_i, rs_i=rs_i)
    for f in range(len(cadus)):
        for p in rng.choice(cadus.shape[1] - 4, int(rng.choice([0, 0, 3, 12, 40])), replace=False):
            cadus[f, 4 + p] ^= int(rng.integers(1, 256))
    sp = synth.SynthSpec(constellation="bpsk", samplerate=3e6, symbolrate=1e6, nrzm=True, seed=9)
    soft = synth.soft_from_symbols(synth.frames_to_symbols(cadus, sp), sp, sigma=15.0, seed=4)
    soft = np.concatenate([soft, rng.integers(-127, 128, 8192).astype(np.int8)])
    soft = soft[: len(soft) // 8192 * 8192]
    cs = cadus.shape[1] * 8
    assert cs < 4096 or rs_i > 1
    want = orc.concat_decode(pyref.fec_cfg(constellation=pyref.BPSK, nrzm=1, rs_usecheck=usecheck, cadu_size=cs, rs_i=rs_i), soft)
    cfg = capi.fec_cfg(constellation="bpsk", nrzm=1, rs_i=rs_i, rs_type=capi.RS223, rs_usecheck=usecheck, cadu_size=cs)
    got, ber, state, st = _run_dev(torch_cuda, capi, cfg, soft)
    assert got.shape == want["cadu"].shape and np.array_equal(got, want["cadu"]) and len(got) >= 12
    if rs_i == 1:  # the overrun is visible: some frame's first byte is not the sync marker's
        assert np.any(got[:, 0] != 0x1A)


@pytest.mark.parametrize("const,sigma,rot", [("qpsk", 30, 1), ("qpsk", 70, 0), ("oqpsk", 40, 1), ("qpsk", 120, 1)])
def test_concat_decoder_qpsk(torch_cuda, capi, orc, const, sigma, rot):
    """JPSS-HRD-like (BASELINE config 4 in miniature): QPSK r=1/2 + NRZ-M + RS, with a 90 degree rotated stream."""
    spec, cadus, plain, syms = util.npp_case(nframes=40, seed=9)
    soft = synth.soft_from_symbols(syms, spec, sigma=sigma, seed=1)
    if rot:
        s2 = soft.copy()
        s2[0::2], s2[1::2] = soft[1::2], -soft[0::2]
        soft = s2
    oc = pyref.fec_cfg(constellation=getattr(pyref, const.upper()), nrzm=1, rs_usecheck=1)
    want = orc.concat_decode(oc, soft)
    cfg = capi.fec_cfg(constellation=const, nrzm=1, rs_i=4, rs_type=capi.RS223, rs_usecheck=1)
    got, ber, state, st = _run_dev(torch_cuda, capi, cfg, soft)
    assert np.array_equal(state, want["state"])
    assert np.array_equal(ber.view(np.uint32), want["ber"].view(np.uint32))
    assert got.shape == want["cadu"].shape and np.array_equal(got, want["cadu"])


@pytest.mark.parametrize("sigma", [20, 50, 80, 130])
def test_metop_decoder(torch_cuda, capi, orc, sigma):
    """MetOp AHRPT (BASELINE configs 1/3 in miniature): QPSK + punctured r=3/4 + RS, deframer SYNCED=18, all frames written."""
    spec, cadus, plain, syms = util.metop_case(nframes=60, seed=5)
    soft = synth.soft_from_symbols(syms, spec, sigma=sigma, seed=2)
    want = orc.metop_decode(soft)
    cfg = capi.fec_cfg(decoder=capi.DEC_METOP_AHRPT, viterbi_ber_thresold=0.17, viterbi_outsync_after=5)
    got, ber, state, st = _run_dev(torch_cuda, capi, cfg, soft)
    assert np.array_equal(state, want["state"])
    assert np.array_equal(ber.view(np.uint32), want["ber"].view(np.uint32))
    assert got.shape == want["cadu"].shape and np.array_equal(got, want["cadu"])


def test_streaming_chunks_equal_one_shot(torch_cuda, capi, orc):
    """Ragged pushes (partial blocks, empty call) give the same CADUs as one call; also the host push/pull path."""
    spec, cadus, plain, syms = util.goes_case(nframes=30, seed=6)
    soft = synth.soft_from_symbols(syms, spec, sigma=28, seed=3)
    oc = pyref.fec_cfg(constellation=pyref.BPSK, nrzm=1, rs_usecheck=1)
    want = orc.concat_decode(oc, soft)["cadu"]
    cfg = capi.fec_cfg(constellation="bpsk", nrzm=1, rs_i=4, rs_type=capi.RS223, rs_usecheck=1)
    n = len(soft)
    bounds = [0, 1000, 1000, 8192 * 3 + 17, 8192 * 11, 8192 * 11 + 5, n]
    got, _, _, _ = _run_dev(torch_cuda, capi, cfg, soft, chunks=bounds)
    nfull = (n // 8192) * 8192
    assert np.array_equal(got, want)
    dec = capi.FecDecoder(cfg)
    for a, b in zip(bounds[:-1], bounds[1:]):
        dec.push(soft[a:b])
    assert np.array_equal(dec.pull(), want)
    assert nfull > 0


def test_sync_loss_and_inversion(torch_cuda, capi, orc):
    """Deframer corner cases: garbage prefix, dropped symbols mid-stream (bit slip), polarity inversion (non-NRZ-M)."""
    spec, cadus, plain, syms = util.goes_case(nframes=36, seed=12)
    spec.nrzm = False
    syms = synth.frames_to_symbols(cadus, spec)
    soft = synth.soft_from_symbols(syms, spec, sigma=20, seed=4)
    rng = np.random.default_rng(1)
    junk = rng.integers(-60, 60, 5000).astype(np.int8)
    cut = 16384 * 9 + 2 * 777  # drop an even number of soft symbols: the Viterbi stays locked, frames slip
    s2 = np.concatenate([junk, soft[:cut], soft[cut + 2 * 1501:16384 * 20], -soft[16384 * 20:]])
    oc = pyref.fec_cfg(constellation=pyref.BPSK, nrzm=0, rs_usecheck=0)
    want = orc.concat_decode(oc, s2)
    cfg = capi.fec_cfg(constellation="bpsk", nrzm=0, rs_i=4, rs_type=capi.RS223, rs_usecheck=0)
    got, ber, state, st = _run_dev(torch_cuda, capi, cfg, s2)
    assert np.array_equal(state, want["state"])
    assert got.shape == want["cadu"].shape and np.array_equal(got, want["cadu"])
    assert len(got) > 25


@pytest.mark.parametrize("name", [c[0] for c in util.SIMPLE_CASES])
@pytest.mark.parametrize("sigma,usecheck", [(15.0, 1), (28.0, 0)])
def test_simple_psk_decoder(torch_cuda, capi, orc, name, sigma, usecheck):
    """ccsds_simple_psk_decoder on the GPU (bit slicers, one or two deframers, derand, RS) == the reference module loop,
    frame for frame and in the reference's output order (deframer_qpsk's frames of a buffer before the main deframer's)."""
    ck, soft, plain = util.simple_case(name, sigma=sigma, nframes=16)
    ock = dict(ck)
    ock["constellation"] = {"bpsk": pyref.BPSK, "qpsk": pyref.QPSK}[ck["constellation"]]
    want = orc.simple_decode(pyref.fec_cfg(decoder=2, rs_usecheck=usecheck, **ock), soft)
    cfg = capi.fec_cfg(decoder=capi.DEC_SIMPLE_PSK, rs_i=4, rs_type=capi.RS223, rs_usecheck=usecheck, **ck)
    got, _, _, st = _run_dev(torch_cuda, capi, cfg, soft)
    assert st.frames_deframed == want["n_deframed"]
    assert got.shape == want["cadu"].shape and np.array_equal(got, want["cadu"])
    # ragged calls (partial buffers, an empty call) and the host push/pull path give the same stream
    n = len(soft)
    bounds = [0, 1001, 1001, 8192 * 2 + 18, 8192 * 5, 8192 * 5 + 6, n]
    got2, _, _, _ = _run_dev(torch_cuda, capi, cfg, soft, chunks=bounds)
    assert np.array_equal(got2, want["cadu"])
    dec = capi.FecDecoder(cfg)
    for a, b in zip(bounds[:-1], bounds[1:]):
        dec.push(soft[a:b])
    assert np.array_equal(dec.pull(), want["cadu"])


@pytest.mark.parametrize("sigma", [15.0, 40.0, 70.0])
def test_viterbi27(torch_cuda, capi, orc, sigma):
    """viterbi::Viterbi27 (viterbi27.cpp: the decoder of the Meteor LRPT / Inmarsat plugin modules = CCDecoder on 2 F soft symbols with an
    erasure tail + MSB-first repack + re-encode BER x 4) as sdhip_op_viterbi27: six chained calls, clean / noisy / hopeless input with
    erased symbols in it -- decoded bytes and the per-call ber() bit-identical to the reference's."""
    import ctypes as C
    from oracle import pyref
    if not pyref.ref_available():
        pytest.skip("needs the compiled reference (viterbi27.cpp)")
    F, nf = 8192, 6
    rng = np.random.default_rng(5)
    tx = synth.conv_encode(rng.integers(0, 2, nf * F, dtype=np.uint8))
    soft = np.clip(np.rint((tx.astype(float) * 2 - 1) * 60 + rng.standard_normal(len(tx)) * sigma), -127, 127).astype(np.int8)
    soft[5] = 0
    soft[100:140] = 0
    want, wber = pyref.ref().viterbi27(F, soft, 1024)
    d_soft = torch_cuda.from_numpy(soft).cuda()
    d_out = torch_cuda.zeros((nf, F // 8), dtype=torch_cuda.uint8, device="cuda")
    ber = np.zeros(nf, dtype=np.float32)
    rc = capi.lib().sdhip_op_viterbi27(0, F, 1024, C.c_void_p(d_soft.data_ptr()), nf, C.c_void_p(d_out.data_ptr()), ber.ctypes.data_as(C.c_void_p))
    assert rc == 0, capi.lib().sdhip_last_error()
    assert np.array_equal(d_out.cpu().numpy(), want) and np.array_equal(ber, wber)
