"""conv_rate != "1/2" of ccsds_conv_concat_decoder (viterbi::Viterbi_Depunc + Depunc23/34/56/78, SURVEY.md 8 row a13') through the C ABI
against the oracle (which is pinned to the compiled reference, tests/test_oracle_vs_ref.py). CADUs, per-block BER and lock state
bit-identical, incl. a loss of lock with re-lock at a stale puncture phase, NRZ-M, the OQPSK IQ-swap search and ragged pushes.
A SYNCED run of calls goes through FecEngine::punc_run (one depuncture launch, one batch of overlapping decoder blocks, one BER
launch; the lock FSM walks the calls afterwards), the lock search and everything around a loss of lock through the call-by-call
path -- the hand-overs between the two are what the loss-of-lock cases exercise. tests/test_fec_gpu_on_twin_cpu.py collects these
tests too (SDHIP_TWIN_FULL=1: minutes on the host twin)."""
import numpy as np
import pytest

from oracle import pyref
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


CASES = [("qpsk", pyref.QPSK, 20.0, 0, False), ("bpsk", pyref.BPSK, 32.0, 1, False), ("oqpsk", pyref.OQPSK, 26.0, 0, True)]


@pytest.mark.parametrize("rate", [1, 2, 3, 4])
@pytest.mark.parametrize("const,oconst,sigma,nrzm,gap", CASES)
def test_punctured_concat_decoder(torch_cuda, capi, orc, rate, const, oconst, sigma, nrzm, gap):
    sigma *= {1: 1.0, 2: 0.85, 3: 0.55, 4: 0.45}[rate]
    soft, plain = util.punctured_case(rate, nframes=8, sigma=sigma, seed=11 + rate, nrzm=bool(nrzm), gap=gap)
    want = orc.concat_decode_punc(pyref.fec_cfg(constellation=oconst, nrzm=nrzm, rs_usecheck=1), rate, soft)
    dec = capi.FecDecoder(capi.fec_cfg(constellation=const, nrzm=nrzm, rs_i=4, rs_type=1, rs_usecheck=1, conv_rate=rate))
    cuts = [0, len(soft) // 3 // 8192 * 8192 + 777, len(soft)]  # ragged pushes: whole blocks are decoded, the rest stays pending
    got = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        dec.push(soft[a:b])
        got.append(dec.pull())
    got = np.concatenate(got)
    assert got.shape == want["cadu"].shape and np.array_equal(got, want["cadu"])
    if not gap:
        assert len(got) >= 6 and sum(i >= 0 for i in util.frame_ids(got, plain)) >= len(got) - 1  # (the 4 ASM bytes are outside the RS code)
    st = dec.stats()
    assert st.viterbi_lock == int(want["state"][-1])
    assert np.float32(st.viterbi_ber) == want["ber"][-1]


@pytest.mark.parametrize("rate,rs_i", [(2, 1), (4, 1), (2, 2)])
def test_punctured_short_cadus_and_the_fill_bytes_overrun(torch_cuda, capi, orc, rate, rs_i):
    """Short CADUs (2048-bit with rs_i = 1, 4096-bit with rs_i = 2) behind a punctured code: a reference call decodes one or TWO windows and
    runs the deframer once over both (module_ccsds_conv_concat_decoder.cpp:93-119), so frames ending in adjacent windows of one call come
    back from one deframer->work() and ReedSolomon's rs_fill_bytes = -1 overrun (reedsolomon.cpp:145-156) writes into the next frame's
    first rs_i bytes -- the engine has to know which windows shared a call. Byte for byte against the oracle."""
    from satdump_amd import synth
    rng = np.random.default_rng(40 + rate + rs_i)
    cadus = synth.make_cadus(40, seed=50 + rate, rs_i=rs_i)
    for f in range(len(cadus)):
        for p in rng.choice(cadus.shape[1] - 4, int(rng.choice([0, 0, 3, 10])), replace=False):
            cadus[f, 4 + p] ^= int(rng.integers(1, 256))
    tx = synth.puncture(synth.conv_encode(np.unpackbits(cadus.reshape(-1))), rate)
    sigma = 20.0 * {2: 0.85, 4: 0.45}[rate]
    soft = np.clip(np.rint((tx.astype(float) * 2 - 1) * 60 + rng.standard_normal(len(tx)) * sigma), -127, 127).astype(np.int8)
    soft = np.concatenate([rng.integers(-127, 128, 333).astype(np.int8), soft, rng.integers(-127, 128, 8192).astype(np.int8)])
    soft = np.concatenate([soft, rng.integers(-127, 128, (-len(soft)) % 8192).astype(np.int8)])
    cs = cadus.shape[1] * 8
    want = orc.concat_decode_punc(pyref.fec_cfg(constellation=pyref.BPSK, nrzm=0, rs_usecheck=0, cadu_size=cs, rs_i=rs_i), rate, soft)["cadu"]
    dec = capi.FecDecoder(capi.fec_cfg(constellation="bpsk", nrzm=0, rs_i=rs_i, rs_type=1, rs_usecheck=0, conv_rate=rate, cadu_size=cs))
    dec.push(soft)
    got = dec.pull()
    assert got.shape == want.shape and np.array_equal(got, want) and len(got) >= 30
    assert np.any(got[:, 0] != 0x1A)  # the overrun is visible: some frame's first byte is not the sync marker's
