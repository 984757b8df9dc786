"""GPU parity against the committed golden vectors (tests/golden/*.npz = what the reference's own code produced,
tests/golden/make_golden.py). Everything goes through the C ABI of libsdhip.so; nothing here needs /root/reference or
the oracle libraries."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def capi():
    from satdump_amd import capi as c
    c.lib()
    return c


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_ccdecoder_golden(torch_cuda, capi):
    d = load("ccdecoder")
    for i in range(5):
        F = int(d[f"c{i}_F"])
        syms = d[f"c{i}_syms"]
        nb = len(syms) // (2 * (F + 6))
        d_s = _dev(torch_cuda, syms)
        d_o = torch_cuda.zeros(nb * F, dtype=torch_cuda.uint8, device="cuda")
        rc = capi.lib().sdhip_op_ccdecoder(0, F, C.c_void_p(d_s.data_ptr()), nb, C.c_void_p(d_o.data_ptr()))
        assert rc == 0, capi.last_error()
        assert np.array_equal(np.packbits(d_o.cpu().numpy()), d[f"c{i}_bits"]), f"case {i}"


@pytest.mark.parametrize("fill,kd,ke", [(-1, "dec", "err"), (0, "dec_fill0", "err_fill0")])
def test_rs_golden(torch_cuda, capi, fill, kd, ke):
    d = load("rs")
    fr = _dev(torch_cuda, d["frames"])
    n = len(d["frames"])
    d_err = torch_cuda.zeros(n * 4, dtype=torch_cuda.int32, device="cuda")
    rc = capi.lib().sdhip_op_rs_decode(0, C.c_void_p(fr.data_ptr() + 4), n, 1024, 1, 4, capi.RS223, fill, C.c_void_p(d_err.data_ptr()))
    assert rc == 0, capi.last_error()
    assert np.array_equal(d_err.cpu().numpy().reshape(n, 4), d[ke])
    assert np.array_equal(fr.cpu().numpy(), d[kd])


def _fec(torch, capi, cfg, soft):
    dec = capi.FecDecoder(cfg)
    d_soft = _dev(torch, soft)
    cap = len(soft) // 4096 + 16
    d_out = torch.zeros((cap, dec.cadu_bytes), dtype=torch.uint8, device="cuda")
    n = dec.process_dev(d_soft.data_ptr(), len(soft), d_out.data_ptr(), cap)
    ber, st = dec.block_taps()
    return d_out[:n].cpu().numpy(), ber, st


@pytest.mark.parametrize("name,const", [("concat_bpsk", "bpsk"), ("concat_qpsk", "qpsk")])
def test_concat_golden(torch_cuda, capi, name, const):
    d = load(name)
    got, ber, st = _fec(torch_cuda, capi, capi.fec_cfg(constellation=const, nrzm=1, rs_i=4, rs_type=capi.RS223, rs_usecheck=1), d["soft"])
    assert np.array_equal(got, d["cadu"])
    assert np.array_equal(st, d["state"]) and np.array_equal(ber.view(np.uint32), d["ber"].view(np.uint32))


def test_metop_golden(torch_cuda, capi):
    d = load("metop")
    got, ber, st = _fec(torch_cuda, capi, capi.fec_cfg(decoder=capi.DEC_METOP_AHRPT, viterbi_ber_thresold=0.17, viterbi_outsync_after=5), d["soft"])
    assert np.array_equal(got, d["cadu"])
    assert np.array_equal(st, d["state"]) and np.array_equal(ber.view(np.uint32), d["ber"].view(np.uint32))


@pytest.mark.parametrize("name,kw", [
    ("demod_goes", dict(samplerate=3e6, symbolrate=927000, constellation="bpsk", rrc_alpha=0.5, pll_bw=0.02, max_sps=3.0)),
    ("demod_metop", dict(samplerate=6e6, symbolrate=2333333, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.003)),
    ("demod_npp", dict(samplerate=30e6, symbolrate=15e6, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.002)),
])
def test_demod_golden_exact_cs16(torch_cuda, capi, name, kw):
    """cs16 samples in (the stored fixture), exact mode: int8 soft symbols AND float symbols bit-identical to the reference's."""
    d = load(name)
    cs16 = d["cs16"]
    n = len(cs16) // 2
    dem = capi.PskDemod(capi.demod_cfg(**kw, exact=1))
    d_x = _dev(torch_cuda, cs16)
    d_soft = torch_cuda.zeros(2 * n + 64, dtype=torch_cuda.int8, device="cuda")
    d_syms = torch_cuda.zeros(2 * (n + 64), dtype=torch_cuda.float32, device="cuda")
    ns = dem.process_dev(d_x.data_ptr(), n, capi.FMT_CS16, d_soft.data_ptr(), 2 * n + 64, d_syms.data_ptr(), n + 64)
    st = dem.stats()
    assert st.buffer_size == int(d["buffer_size"]) and np.float32(st.final_sps) == d["final_sps"]
    assert ns == len(d["soft"])
    assert np.array_equal(d_soft[:ns].cpu().numpy(), d["soft"])
    nsym = len(d["syms"])
    assert np.array_equal(d_syms[: 2 * nsym].cpu().numpy().view(np.uint32), d["syms"].view(np.uint32).reshape(-1))


def test_demod_carrier_golden_exact_cs16(torch_cuda, capi):
    """psk_demod with has_carrier (carrier-tracking PLL + DC block in front of the Costas loop), cs16 fixture, exact mode: int8 soft
    symbols AND float symbols bit-identical to the reference's."""
    from tests.test_zy_demod_additions_gpu import _carrier_case
    d = load("demod_carrier")
    cs16 = d["cs16"]
    n = len(cs16) // 2
    _, kw = _carrier_case(nframes=1)
    dem = capi.PskDemod(capi.demod_cfg(constellation="bpsk", exact=1, **kw))
    d_x = _dev(torch_cuda, cs16)
    d_soft = torch_cuda.zeros(2 * n + 64, dtype=torch_cuda.int8, device="cuda")
    d_syms = torch_cuda.zeros(2 * (n + 64), dtype=torch_cuda.float32, device="cuda")
    ns = dem.process_dev(d_x.data_ptr(), n, capi.FMT_CS16, d_soft.data_ptr(), 2 * n + 64, d_syms.data_ptr(), n + 64)
    assert ns == len(d["soft"]) and np.array_equal(d_soft[:ns].cpu().numpy(), d["soft"])
    nsym = len(d["syms"])
    assert np.array_equal(d_syms[: 2 * nsym].cpu().numpy().view(np.uint32), d["syms"].view(np.uint32).reshape(-1))


@pytest.mark.parametrize("name", ["bpsk_nrzm", "qpsk_diff_swap", "qpsk_90deg"])
def test_simple_decoder_golden(torch_cuda, capi, name):
    from tests import util
    d = load("simple_" + name)
    ck = dict(next(c for c in util.SIMPLE_CASES if c[0] == name)[1])
    got, _, _ = _fec(torch_cuda, capi, capi.fec_cfg(decoder=capi.DEC_SIMPLE_PSK, rs_i=4, rs_type=capi.RS223, rs_usecheck=0, **ck), d["soft"])
    assert np.array_equal(got, d["cadu"])


def test_gardner_golden(torch_cuda, capi):
    d = load("gardner")
    x = (d["cs16"].astype(np.float32) * np.float32(1.0 / 32767.0))
    n = len(x) // 2
    d_x = _dev(torch_cuda, x)
    d_y = torch_cuda.zeros(2 * (n + 64), dtype=torch_cuda.float32, device="cuda")
    p = np.asarray(d["params"], dtype=np.float32)
    nout = capi.lib().sdhip_op_block(0, 7, p.ctypes.data_as(C.c_void_p), C.c_void_p(d_x.data_ptr()), n, C.c_void_p(d_y.data_ptr()), n + 64)
    assert nout == len(d["syms"]), capi.last_error()
    assert np.array_equal(d_y[: 2 * nout].cpu().numpy().view(np.uint32), d["syms"].view(np.uint32).reshape(-1))
