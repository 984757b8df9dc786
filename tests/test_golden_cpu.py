"""The plain-C restatement (oracle/sd_oracle.c) against the committed golden vectors (tests/golden/*.npz = outputs of the
reference's own code, tests/golden/make_golden.py). Runs without a GPU and without /root/reference: this is what pins the
oracle where the compiled reference cannot travel."""
import os

import numpy as np
import pytest

from oracle import pyref

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def test_index_lists_every_fixture():
    idx = open(os.path.join(G, "INDEX.txt")).read()
    for f in os.listdir(G):
        if f.endswith(".npz"):
            assert f in idx


def test_ccdecoder_golden(port):
    d = load("ccdecoder")
    for i in range(5):
        F = int(d[f"c{i}_F"])
        bits = port.ccdecoder(F, d[f"c{i}_syms"])
        assert np.array_equal(np.packbits(bits), d[f"c{i}_bits"]), f"case {i}"


def test_rs_golden(port):
    d = load("rs")
    for fill, kd, ke in ((-1, "dec", "err"), (0, "dec_fill0", "err_fill0")):
        dec, err = port.rs_decode(d["frames"], fill_bytes=fill)
        assert np.array_equal(err, d[ke]) and np.array_equal(dec, d[kd])
    assert (d["err"] == -1).any() and (d["err"] == 16).any()


@pytest.mark.parametrize("name,const", [("concat_bpsk", pyref.BPSK), ("concat_qpsk", pyref.QPSK)])
def test_concat_golden(port, name, const):
    d = load(name)
    r = port.concat_decode(pyref.fec_cfg(constellation=const, nrzm=1, rs_usecheck=1), d["soft"])
    assert np.array_equal(r["cadu"], d["cadu"]) and len(d["cadu"]) >= 10
    assert np.array_equal(r["state"], d["state"]) and np.array_equal(r["ber"].view(np.uint32), d["ber"].view(np.uint32))
    assert np.array_equal(r["frm_err"], d["frm_err"])


def test_metop_golden(port):
    d = load("metop")
    r = port.metop_decode(d["soft"])
    assert np.array_equal(r["cadu"], d["cadu"]) and len(d["cadu"]) >= 6
    assert np.array_equal(r["state"], d["state"]) and np.array_equal(r["ber"].view(np.uint32), d["ber"].view(np.uint32))
    assert np.array_equal(r["frm_err"], d["frm_err"])


@pytest.mark.parametrize("name,cfg", [
    ("demod_goes", dict(samplerate=3e6, symbolrate=927000, constellation=pyref.BPSK, pll_bw=0.02, max_sps=3.0)),
    ("demod_metop", dict()),
    ("demod_npp", dict(samplerate=30e6, symbolrate=15e6, constellation=pyref.QPSK, pll_bw=0.002)),
])
def test_demod_golden(port, name, cfg):
    d = load(name)
    x = (d["cs16"].astype(np.float32) * np.float32(1.0 / 32767.0)).view(np.complex64)
    r = port.psk_demod(pyref.demod_cfg(**cfg), x)
    assert r["buffer_size"] == int(d["buffer_size"]) and np.float32(r["final_sps"]) == d["final_sps"]
    assert np.array_equal(r["soft"], d["soft"])
    assert np.array_equal(r["syms"].view(np.uint32), d["syms"].view(np.uint32))  # float symbols bit for bit


def test_demod_carrier_golden(port):
    """psk_demod's has_carrier chain and the carrier PLL block alone (pll_carrier_tracking.cpp, fast_trig.cpp), cs16 fixture."""
    from tests.test_zy_demod_additions_gpu import _carrier_case
    d = load("demod_carrier")
    x = (d["cs16"].astype(np.float32) * np.float32(1.0 / 32767.0)).view(np.complex64)
    _, kw = _carrier_case(nframes=1)
    r = port.psk_demod(pyref.demod_cfg(constellation=pyref.BPSK, **kw), x)
    assert np.array_equal(r["soft"], d["soft"]) and np.array_equal(r["syms"].view(np.uint32), d["syms"].view(np.uint32))
    got = port.block(8, d["pll_params"], x[:40000])
    assert np.array_equal(got.view(np.uint32), d["pll_out"].view(np.uint32))


def test_taps_golden(port):
    d = load("taps")
    assert np.array_equal(port.rrc_taps(2.7e6, 927000, 0.5, 31).view(np.uint32), d["rrc_goes"].view(np.uint32))
    assert np.array_equal(port.rrc_taps(6e6, 2333333, 0.5, 31).view(np.uint32), d["rrc_metop"].view(np.uint32))
    assert np.array_equal(port.mm_bank(128, 8).view(np.uint32), d["mm"].view(np.uint32))
    bank, ir, dr = port.resamp_bank(2700000, 3000000)
    assert [ir, dr] == list(d["resamp_ratio"]) and np.array_equal(bank.view(np.uint32), d["resamp"].view(np.uint32))


@pytest.mark.parametrize("name", ["bpsk_nrzm", "qpsk_diff_swap", "qpsk_90deg"])
def test_simple_decoder_golden(port, name):
    from tests import util
    d = load("simple_" + name)
    ck = dict(next(c for c in util.SIMPLE_CASES if c[0] == name)[1])
    ck["constellation"] = {"bpsk": pyref.BPSK, "qpsk": pyref.QPSK}[ck["constellation"]]
    r = port.simple_decode(pyref.fec_cfg(decoder=2, rs_usecheck=0, **ck), d["soft"])
    assert np.array_equal(r["cadu"], d["cadu"]) and len(d["cadu"]) >= 6
    assert np.array_equal(r["frm_err"], d["frm_err"])


def test_gardner_golden(port):
    d = load("gardner")
    x = (d["cs16"].astype(np.float32) * np.float32(1.0 / 32767.0)).view(np.complex64)
    got = port.block(7, d["params"], x)
    assert np.array_equal(got.view(np.uint32), d["syms"].view(np.uint32)) and len(got) > 10000


@pytest.mark.parametrize("name", ["punct_r34", "punct_r78"])
def test_punctured_decoder_golden(port, name):
    g = load(name)
    r = port.concat_decode_punc(pyref.fec_cfg(constellation=pyref.QPSK, nrzm=0, rs_usecheck=1), int(g["rate"]), g["soft"])
    assert len(g["cadu"]) >= 4
    for k in ("cadu", "ber", "state", "frm_err"):
        assert np.array_equal(r[k], g[k]), k
