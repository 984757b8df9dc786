"""tests/test_dvbs2_gpu.py's parity cases against the HOST TWIN of dvbs2_ldpc.hip (tests/emu) -- a subset sized for the CPU suite --
plus what needs no device: the generated tables are what the generator makes of the reference's header, and the oracle itself
round-trips (its encoder's code words decode to themselves)."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyref
from tests import test_dvbs2_gpu as G
from tests.emu import build as emu_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    if not os.path.exists(emu_build.CLANG):
        pytest.skip("no host clang++ to build the twin with")
    lib = emu_build.build()
    spec = importlib.util.spec_from_file_location("capi_host_twin4", os.path.join(ROOT, "satdump_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    old = os.environ.get("SDHIP_LIB")
    os.environ["SDHIP_LIB"] = lib
    os.environ["SDHIP_TESTING_TWIN"] = "1"
    try:
        spec.loader.exec_module(m)
        m.lib()
    finally:
        del os.environ["SDHIP_TESTING_TWIN"]
        if old is None:
            del os.environ["SDHIP_LIB"]
        else:
            os.environ["SDHIP_LIB"] = old
    return m


@pytest.mark.parametrize("fs,rate", [(0, "2/3"), (0, "9/10"), (0, "1/4"), (1, "1/2"), (1, "3/5"), (1, "8/9")])
def test_codes_bit_exact_on_the_twin(capi, fs, rate):
    ref = G._ref(False)
    G.run_case(capi, ref, fs, rate, 2, 20, G.SIGMA[rate] * (1.0 if fs == 0 else 0.9), 8)
    G.run_case(capi, ref, fs, rate, 1, 60, 1.0, 5, seed=2)


def test_sse_batch_on_the_twin(capi):
    ref = G._ref(True)
    rc = capi.S2_RATES["1/2"]
    soft, bits, k = G.make_frames(ref, 1, rc, 16, 20, 16.0, 5)
    clean, _, _ = G.make_frames(ref, 1, rc, 16, 60, 1.0, 5)
    soft[2:] = clean[2:]
    want, wt = ref.ldpc_decode(1, rc, soft, 10)
    dec = capi.LdpcDecoder(framesize=1, rate="1/2", batch=16)
    got = soft.copy()
    gt = dec.decode(got, 10)
    assert np.array_equal(gt, wt) and np.array_equal(got, want)


@pytest.mark.parametrize("fs,rate", [(0, "2/3"), (0, "1/2"), (0, "9/10"), (1, "1/4"), (1, "8/9")])
def test_bch_bit_exact_on_the_twin(capi, fs, rate):
    G.bch_case(capi, G._ref(False), fs, rate, [0, 1, 2, 3, 5, 8, 10, 12, 13, 14, 20, 40])


@pytest.mark.parametrize("fs,rate", [(0, "2/3"), (1, "1/2")])
def test_bb_descrambler_on_the_twin(capi, fs, rate):
    """sdhip_bb_descramble_dev == BBFrameDescrambler::work (the twin's "device" pointers are host pointers)."""
    ref = G._ref(False)
    rc = capi.S2_RATES[rate]
    dec = capi.BchDecoder(framesize=fs, rate=rate)
    fr = np.random.default_rng(2).integers(0, 256, (5, dec.nbch // 8 + 3), dtype=np.uint8)
    want = ref.bb_descramble(fs, rc, fr)
    got = fr.copy()
    dec.descramble_dev(got.ctypes.data, len(got), got.shape[1])
    assert np.array_equal(got, want) and not np.array_equal(got, fr) and np.array_equal(got[:, dec.kbch // 8:], fr[:, dec.kbch // 8:])


@pytest.mark.parametrize("const,fs,rate", [(0, 0, 5), (1, 0, 5), (1, 0, 4), (1, 1, 4), (2, 0, 6), (3, 1, 7)])
def test_s2_deinterleaver_on_the_twin(capi, const, fs, rate):
    """sdhip_s2_deinterleave_dev == S2Deinterleaver::deinterleave for QPSK / 8PSK (incl. the reversed columns of rate 3/5) / 16APSK / 32APSK."""
    import ctypes as C
    ref = G._ref(False)
    n = 64800 if fs == 0 else 16200
    x = np.random.default_rng(const + rate).integers(-128, 128, (3, n), dtype=np.int8)
    want = ref.s2_deinterleave(const, fs, rate, x)
    got = np.zeros_like(x)
    rc = capi.lib().sdhip_s2_deinterleave_dev(0, const, fs, rate, x.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), 3)
    assert rc == 0 and np.array_equal(got, want)


@pytest.mark.parametrize("modcod,short,pilots", [(6, 0, 0), (13, 1, 0), (20, 0, 0), (6, 1, 1), (13, 0, 1)])
def test_bb_to_soft_on_the_twin(capi, modcod, short, pilots):
    def to_dev(a):
        a = np.ascontiguousarray(a)
        return (a, a.ctypes.data)

    def zeros_dev(n, dt):
        a = np.zeros(n, dtype=dt)
        return (a, a.ctypes.data)

    G.check_bb_to_soft(capi, to_dev, lambda d: d[0], zeros_dev, modcod, short, pilots, nframes=3)


@pytest.mark.parametrize("slots,pilots,kind", [(90, 0, "slips"), (60, 1, "slips"), (45, 0, "locked"), (90, 0, "noise")])
def test_pl_sync_on_the_twin(capi, slots, pilots, kind):
    def to_dev(a):
        a = np.ascontiguousarray(a)
        return (a, a.ctypes.data)

    def zeros_dev(n, dt):
        a = np.zeros(n, dtype=dt)
        return (a, a.ctypes.data)

    G.check_pl_sync(capi, to_dev, lambda d: d[0], zeros_dev, slots, pilots, kind)


def _np_helpers():
    def to_dev(a):
        a = np.ascontiguousarray(a)
        return (a, a.ctypes.data)

    def zeros_dev(n, dt):
        a = np.zeros(n, dtype=dt)
        return (a, a.ctypes.data)

    return to_dev, (lambda d: d[0]), zeros_dev


def test_atan2f_on_the_twin(capi):
    G.check_atan2f(capi, *_np_helpers())


@pytest.mark.parametrize("modcod,short,pilots", [(4, 1, 0), (13, 1, 0), (6, 1, 1)])
def test_pll_on_the_twin(capi, modcod, short, pilots):
    G.check_pll(capi, *_np_helpers(), modcod, short, pilots, nfr=3)


def test_symbols_to_bbframes_on_the_twin(capi):
    G.check_symbols_to_bbframes(capi, *_np_helpers(), nfr=3)


@pytest.mark.parametrize("front", ["exact", "chunk"])
def test_baseband_to_bbframes_on_the_twin(capi, front):
    G.check_symbols_to_bbframes(capi, *_np_helpers(), nfr=7, via_baseband=front)


class _NumpyMem:
    """satdump_amd.dvbs2's memory adapter on host arrays (the twin's 'device' pointers are host pointers)."""

    @staticmethod
    def alloc(n, dtype):
        return np.zeros(int(n), dtype=dtype)

    @staticmethod
    def from_host(a):
        return np.ascontiguousarray(a).copy()

    @staticmethod
    def ptr(h):
        return h.ctypes.data

    @staticmethod
    def to_host(h, n=None):
        return (h if n is None else h[:n]).copy()


def test_dvbs2_demod_mirror_on_the_twin(capi):
    G.check_dvbs2_demod_mirror(capi, _NumpyMem)


@pytest.mark.parametrize("modcod,short,esn0_db,nfr,lane_len", [(12, 1, 9.0, 12, 0), (12, 1, 9.0, 8, 1500), (6, 1, 6.0, 8, 0)])
def test_pll_parallel_on_the_twin(capi, modcod, short, esn0_db, nfr, lane_len):
    G.check_pll_parallel(capi, *_np_helpers(), modcod, short, esn0_db, nfr, lane_len)


def test_dvbs2_engine_parallel_on_the_twin(capi):
    st = G.check_dvbs2_engine(capi, _NumpyMem, modcod=12, short=1, nfr=10, esn0_db=10.0, acq=3 * 5490)
    assert st["pll_serial_frames"] == 3


def test_dvbs2_engine_ragged_calls_on_the_twin(capi):
    G.check_dvbs2_engine(capi, _NumpyMem, modcod=12, short=1, nfr=10, esn0_db=10.0, cuts=[0, 40001, 40002, 83457, 10 * 5490 * 2 + 3 * 5490 * 2], acq=3 * 5490)


def test_dvbs2_engine_freq_prop_on_the_twin(capi):
    G.check_dvbs2_engine(capi, _NumpyMem, modcod=4, short=1, nfr=20, esn0_db=7.0, freq_prop=0.08, acq=3 * 8190, cfo_hz=50.0,
                         cuts=[0] + [8190 * 2 * 5 * k for k in range(1, 5)] + [23 * 8190 * 2])


@pytest.mark.skipif(not os.environ.get("SDHIP_TWIN_FULL"), reason="two minutes on the emulated kernels beside test_dvbs2_engine_freq_prop_on_the_twin; the GPU suite runs it (SDHIP_TWIN_FULL=1 runs it here)")
def test_dvbs2_engine_freq_prop_hand_over_on_the_twin(capi):
    """the symbols waiting in the PL synchroniser's ring at a hand-over are turned on at the new rate (they reached the loop with a +d / -d frequency
    step before). A larger offset, the module's default factor, a hand-over every two frames: every frame the reference chain finds (without the feedback) comes
    out, in order."""
    G.check_dvbs2_engine(capi, _NumpyMem, modcod=4, short=1, nfr=20, esn0_db=7.0, freq_prop=0.01, acq=3 * 8190, cfo_hz=120.0,
                         cuts=[0] + [8190 * 2 * 2 * k for k in range(1, 11)] + [23 * 8190 * 2], min_handover=0.1)


def test_bb_to_soft_golden_on_the_twin(capi):
    def to_dev(a):
        a = np.ascontiguousarray(a)
        return (a, a.ctypes.data)

    def zeros_dev(n, dt):
        a = np.zeros(n, dtype=dt)
        return (a, a.ctypes.data)

    G.check_bb_to_soft_golden(capi, to_dev, lambda d: d[0], zeros_dev)


test_bb_to_soft_refusals = G.test_bb_to_soft_refusals
test_errors = G.test_errors


def test_generated_tables_are_the_generators_output():
    if not os.path.isdir("/root/reference/plugins/dvb_support"):
        pytest.skip("reference tree not present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_dvbs2_tables.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "satdump_amd", "csrc", "dvbs2_tables.inc")).read()


@pytest.mark.parametrize("sse", [False, True])
def test_oracle_round_trip(sse):
    """The checker checks out: a code word of every table (tests/dvbs2_util.py) is a fixed point of the reference's decoder (0 update
    passes), in both builds -- which also pins the generated tables and the test encoder against the reference's graph."""
    ref = G._ref(sse)
    for fs, rate in [(0, r) for r in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11)] + [(1, r) for r in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10)]:
        soft, bits, k = G.make_frames(ref, fs, rate, ref.batch, 40, 0.0, 3)
        out, tr = ref.ldpc_decode(fs, rate, soft, 5)
        assert tr.tolist() == [0] and np.array_equal(out, soft)


@pytest.mark.parametrize("name,kw", G.TS_CASES)
def test_s2_ts_extractor_on_the_twin(capi, name, kw):
    G.test_s2_ts_extractor(capi, name, kw)
