"""fengyun_ahrpt_decoder on the device (the FY-3 decoder of plugins/fengyun3_support; SDHIP_DEC_FENGYUN_AHRPT of the FEC handle,
satdump_amd/csrc/fec_engine.hip process_blocks_fengyun) against the reference module's own loop on the reference's own classes (oracle/ref_wrap.cpp:
sdref_fy3_decode -- rotate_soft, two Viterbi3_4 in fymode, FengyunDiff::work2, BPSK_CCSDS_Deframer with the module's thresholds, derand_ccsds, ReedSolomon,
compiled in place): byte work, bit-exact CADUs, and every read's BER figure and lock state of both Viterbis -- clean and noisy streams, the second rail
complemented or not, the differential decoder's inputs the other way round (the module's noSyncRuns counter exchanges them), a stream that starts on
the other symbol of a pair (its viterbiNoSyncRun counter moves `shift`, after which every read loses a pair), bytes missing in the middle, noise, any
cut of the stream into calls, the host push / pull path."""
import numpy as np
import pytest

from oracle import pyref
from satdump_amd import synth
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import torch
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")
    from satdump_amd import capi as c
    c.lib()
    return c


def _torch_helpers():
    import torch

    def to_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return (t, t.data_ptr())

    def zeros_dev(n, dt):
        t = torch.zeros(n, dtype={np.int8: torch.int8, np.uint8: torch.uint8}[dt], device="cuda")
        return (t, t.data_ptr())

    return to_dev, (lambda d: d[0].cpu().numpy()), zeros_dev


def run_engine(capi, to_dev, to_host, zeros_dev, soft, invert_second=True, cuts=None, ber_thr=0.17, outsync_after=5, mpt=False):
    cfg = capi.fec_cfg(decoder=capi.DEC_FENGYUN_MPT if mpt else capi.DEC_FENGYUN_AHRPT, viterbi_ber_thresold=ber_thr, viterbi_outsync_after=outsync_after,
                       invert_second_viterbi=int(invert_second))
    dec = capi.FecDecoder(cfg)
    outs, bers, states = [], [], []
    cuts = cuts or [0, len(soft)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        d_in = to_dev(soft[a:b]) if b > a else zeros_dev(16, np.int8)
        cap = (b - a) // 8192 + 8
        d_out = zeros_dev(cap * 1024, np.uint8)
        n = dec.process_dev(d_in[1], b - a, d_out[1], cap)
        outs.append(to_host(d_out)[: n * 1024].reshape(n, 1024).copy())
        be, st = dec.block_taps()
        bers.append(be)
        states.append(st)
    return np.concatenate(outs, axis=0), np.concatenate(bers).reshape(-1, 2), np.concatenate(states).reshape(-1, 2), dec.stats()


CASES = [
    dict(nframes=24),
    dict(nframes=24, invert_second=False),
    dict(nframes=30, sigma=26.0),
    dict(nframes=30, sigma=31.0),                        # around the lock threshold: searches, drops, re-locks
    dict(nframes=48, branches_swapped=True),             # ten reads of a deframer without sync, then the inputs are exchanged
    dict(nframes=48, lead=16384 * 2 + 2),                # the other symbol of a pair first: `shift` moves after ten reads without lock
    dict(nframes=40, gaps=((16384 * 9 + 4 * 333, 4 * 777),)),   # whole puncture periods missing: the Viterbis keep their lock, the frames slip
    dict(nframes=40, gaps=((16384 * 9 + 4 * 333, 4 * 777 + 2),), sigma=12.0),  # one symbol per rail missing: both Viterbis lose lock and search again
    dict(nframes=12, lead=16384 * 3, noise_tail=16384 * 4),
    dict(nframes=0, noise_tail=16384 * 14),               # noise only: shift toggles on every read from the tenth on
]


def check_decoder(capi, to_dev, to_host, zeros_dev, case, cuts=None):
    kw = dict(case)
    inv2 = kw.get("invert_second", True)
    mpt = kw.get("mpt", False)
    soft, plain = synth.fy3_ahrpt_soft(**kw) if kw.get("nframes") else (np.random.default_rng(5).integers(-60, 60, kw["noise_tail"]).astype(np.int8), None)
    want = pyref.ref().fy3_mpt_decode(soft) if mpt else pyref.ref().fy3_decode(soft, invert_second=inv2)
    got, ber, state, st = run_engine(capi, to_dev, to_host, zeros_dev, soft, invert_second=inv2, cuts=cuts, mpt=mpt)
    assert np.array_equal(state, want["state"])
    assert np.array_equal(ber.view(np.uint32), want["ber"].view(np.uint32))
    assert got.shape == want["cadu"].shape and np.array_equal(got, want["cadu"])
    if len(want["cadu"]):
        assert list(st.rs_errors[:4]) == list(want["frm_err"][-1])
    assert st.viterbi_lock == want["state"][-1, 0] and st.viterbi2_lock == want["state"][-1, 1]
    if kw.get("nframes", 0) >= 24 and kw.get("sigma", 18.0) <= 26.0 and not kw.get("gaps"):
        ids = util.frame_ids(got, plain)
        assert sum(i >= 0 for i in ids) >= kw["nframes"] // 2  # the stream itself was decodable: frames of the transmitted list come out
    return want


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_fy3_decoder(capi, case):
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_fy3_decode")):
        pytest.skip("needs the compiled reference")
    check_decoder(capi, *_torch_helpers(), case)


def check_cuts(capi, to_dev, to_host, zeros_dev, nframes=36):
    """Any cut of the stream into calls (partial reads, an empty call) gives the one-call output; so does the host push / pull path."""
    soft, _ = synth.fy3_ahrpt_soft(nframes, seed=9, sigma=22.0)
    want = pyref.ref().fy3_decode(soft)
    n = len(soft)
    cuts = [0, 1000, 1000, 16384 * 3 + 17, 16384 * 11, 16384 * 11 + 5, n]
    got, ber, state, _ = run_engine(capi, to_dev, to_host, zeros_dev, soft, cuts=cuts)
    assert np.array_equal(got, want["cadu"])
    assert np.array_equal(state, want["state"]) and np.array_equal(ber.view(np.uint32), want["ber"].view(np.uint32))
    dec = capi.FecDecoder(capi.fec_cfg(decoder=capi.DEC_FENGYUN_AHRPT, viterbi_ber_thresold=0.17, viterbi_outsync_after=5, invert_second_viterbi=1))
    for a, b in zip(cuts[:-1], cuts[1:]):
        dec.push(soft[a:b])
    assert np.array_equal(dec.pull(), want["cadu"])


def test_fy3_cuts_and_host_path(capi):
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_fy3_decode")):
        pytest.skip("needs the compiled reference")
    check_cuts(capi, *_torch_helpers())


def test_fy3_long_run(capi):
    """A run long enough for the batched path to carry most of it (several hundred reads in one call), with a stretch of noise in the middle."""
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_fy3_decode")):
        pytest.skip("needs the compiled reference")
    a, _ = synth.fy3_ahrpt_soft(300, seed=21, sigma=20.0)
    b, _ = synth.fy3_ahrpt_soft(300, seed=22, sigma=20.0, branches_swapped=True)
    noise = np.random.default_rng(8).integers(-60, 60, 16384 * 13 + 6).astype(np.int8)
    soft = np.concatenate([a, noise, b])
    want = pyref.ref().fy3_decode(soft)
    got, ber, state, st = run_engine(capi, *_torch_helpers(), soft)
    assert np.array_equal(state, want["state"]) and np.array_equal(ber.view(np.uint32), want["ber"].view(np.uint32))
    assert got.shape == want["cadu"].shape and np.array_equal(got, want["cadu"])
    assert len(got) > 500


# fengyun_mpt_decoder (plugins/fengyun3_support/fengyun3/module_fengyun_mpt_decoder.cpp; SDHIP_DEC_FENGYUN_MPT): the same loop on two Viterbi1_2 -- rate-1/2 rails,
# phases 0 / 90 searched, the rails' byte pairs exchanged in front of the decoders, the deframer's default thresholds, the watchdog on Viterbi 1's state alone --
# against the module's loop on the reference's own classes (oracle/ref_wrap.cpp: sdref_fy3_mpt_decode)
MPT_CASES = [
    dict(nframes=24, mpt=True),
    dict(nframes=30, sigma=30.0, mpt=True),
    dict(nframes=30, sigma=40.0, mpt=True),                        # around the lock threshold
    dict(nframes=48, branches_swapped=True, mpt=True),
    dict(nframes=48, lead=16384 * 2 + 2, mpt=True),                # the other symbol of a pair first
    dict(nframes=40, gaps=((16384 * 9 + 4 * 333, 4 * 777 + 2),), sigma=12.0, mpt=True),
    dict(nframes=12, lead=16384 * 3, noise_tail=16384 * 4, mpt=True),
]


@pytest.mark.parametrize("case", MPT_CASES, ids=[str(i) for i in range(len(MPT_CASES))])
def test_fy3_mpt_decoder(capi, case):
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_fy3_mpt_decode")):
        pytest.skip("needs the compiled reference")
    check_decoder(capi, *_torch_helpers(), case)


def check_mpt_cuts(capi, to_dev, to_host, zeros_dev, nframes=36):
    soft, _ = synth.fy3_ahrpt_soft(nframes, seed=9, sigma=24.0, mpt=True)
    want = pyref.ref().fy3_mpt_decode(soft)
    n = len(soft)
    cuts = [0, 1000, 1000, 16384 * 3 + 17, 16384 * 11, 16384 * 11 + 5, n]
    got, ber, state, _ = run_engine(capi, to_dev, to_host, zeros_dev, soft, cuts=cuts, mpt=True)
    assert np.array_equal(got, want["cadu"]) and len(got) >= nframes - 6
    assert np.array_equal(state, want["state"]) and np.array_equal(ber.view(np.uint32), want["ber"].view(np.uint32))
    dec = capi.FecDecoder(capi.fec_cfg(decoder=capi.DEC_FENGYUN_MPT, viterbi_ber_thresold=0.17, viterbi_outsync_after=5))
    for a, b in zip(cuts[:-1], cuts[1:]):
        dec.push(soft[a:b])
    assert np.array_equal(dec.pull(), want["cadu"])


def test_fy3_mpt_cuts_and_host_path(capi):
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_fy3_mpt_decode")):
        pytest.skip("needs the compiled reference")
    check_mpt_cuts(capi, *_torch_helpers())
