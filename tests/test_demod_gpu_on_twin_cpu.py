"""The GPU parity tests of the demodulator (tests/test_demod_gpu.py), the very same test functions, collected a second time against
the HOST TWIN of the engine (tests/emu): `torch_cuda` is a numpy stand-in, `capi` the ctypes binding opened on the twin, and the
FEC behind it (the twin has none) is the oracle's. Runs in the CPU suite (-m "not gpu"); proves host logic and arithmetic, not the
GPU build -- that stays with -m gpu."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import pyref
from tests import test_demod_gpu as G
from tests import test_zy_demod_additions_gpu as G2
from tests.emu import build as emu_build
from tests.emu import fake_torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleFec:
    """capi.FecDecoder's push/pull on the oracle (the host twin covers the demodulator only)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.soft = []

    def push(self, soft):
        self.soft.append(np.ascontiguousarray(soft, dtype=np.int8))

    def pull(self, max_frames=1 << 20):
        s = np.concatenate(self.soft)
        o = pyref.best()
        if self.cfg.decoder == 1:
            return o.metop_decode(s, ber_thr=self.cfg.viterbi_ber_thresold, outsync_after=self.cfg.viterbi_outsync_after)["cadu"]
        oc = pyref.fec_cfg(**{f: getattr(self.cfg, f) for f, _ in self.cfg._fields_})
        return (o.simple_decode(oc, s) if self.cfg.decoder == 2 else o.concat_decode(oc, s))["cadu"]


class _TwinCapi:
    def __init__(self, mod):
        self._m = mod

    def __getattr__(self, k):
        if k == "FecDecoder":
            return _OracleFec
        if k == "fec_cfg":  # the defaults of the real library (a host function; the twin's FEC entries are stubs)
            from satdump_amd import capi as real
            return real.fec_cfg
        return getattr(self._m, k)


@pytest.fixture(scope="module")
def torch_cuda():
    return fake_torch


@pytest.fixture(scope="module")
def capi():
    if not os.path.exists(emu_build.CLANG):
        pytest.skip("no host clang++ to build the twin with")
    lib = emu_build.build()
    spec = importlib.util.spec_from_file_location("capi_host_twin2", os.path.join(ROOT, "satdump_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    old = os.environ.get("SDHIP_LIB")
    os.environ["SDHIP_LIB"] = lib
    os.environ["SDHIP_TESTING_TWIN"] = "1"  # capi refuses the twin without it
    try:
        spec.loader.exec_module(m)
        m.lib()
    finally:
        del os.environ["SDHIP_TESTING_TWIN"]
        if old is None:
            del os.environ["SDHIP_LIB"]
        else:
            os.environ["SDHIP_LIB"] = old
    assert m.LIB_PATH == lib  # the twin, not lib/libsdhip.so
    return _TwinCapi(m)


@pytest.fixture(scope="module")
def orc():
    return pyref.best()


# the same functions, parametrisation included; the 2 GB round trip stays a GPU-only test
test_single_blocks_bit_exact = G.test_single_blocks_bit_exact
test_sincos_matches_host_libm = G.test_sincos_matches_host_libm
test_exact_mode_bit_identical = G.test_exact_mode_bit_identical
test_exact_mode_streaming = G.test_exact_mode_streaming
test_chunked_mode_symbols_and_cadus = G.test_chunked_mode_symbols_and_cadus
test_chunked_mode_streaming_calls = G.test_chunked_mode_streaming_calls
test_every_symbol_beyond_tolerance_is_an_arm_flip = G.test_every_symbol_beyond_tolerance_is_an_arm_flip
test_cs16_input = G.test_cs16_input
test_empty_and_tiny_calls = G.test_empty_and_tiny_calls
test_integer_input_formats = G.test_integer_input_formats
test_exact_mode_other_constellations = G.test_exact_mode_other_constellations


def test_noise_only_input_is_bounded(torch_cuda, capi):
    """tests/test_demod_gpu.py::test_noise_only_input_is_bounded without its wall-clock bound (the twin is not a timing model)."""
    n = 1_000_000
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(2 * n) * 0.3).astype(np.float32)
    dem = capi.PskDemod(capi.demod_cfg(samplerate=6e6, symbolrate=2333333, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.003))
    soft = np.zeros(2 * n + 64, dtype=np.int8)
    for _ in range(2):
        ns = dem.process_dev(x.ctypes.data, n, capi.FMT_CF32, soft.ctypes.data, 2 * n + 64)
    st = dem.stats()
    assert abs(ns / 2 - n / (6e6 / 2333333)) < 0.01 * n
    assert st.chunks_forced > 0 and st.chunks_fixed < 6 * st.chunks


test_host_push_pull_path = G.test_host_push_pull_path
test_dc_block_in_front_of_the_resampler = G2.test_dc_block_in_front_of_the_resampler
test_dc_block_chunk_parallel = G2.test_dc_block_chunk_parallel
test_power_of_two_predecimator = G2.test_power_of_two_predecimator
test_custom_samplerate = G2.test_custom_samplerate
test_post_costas_dc = G2.test_post_costas_dc
test_has_carrier = G2.test_has_carrier
test_soft_symbols_without_the_float_symbols = G2.test_soft_symbols_without_the_float_symbols
test_freq_shift = G2.test_freq_shift
test_doppler = G2.test_doppler
test_dvbs2_front_end = G2.test_dvbs2_front_end
test_cooperative_lanes_equal_the_per_lane_streams = G2.test_cooperative_lanes_equal_the_per_lane_streams



@pytest.mark.parametrize("case,esn0,max_diff", [("metop", 5.5, 0.005), ("npp", 3.0, 0.005)])
def test_margin_sweep_cadu_identity(torch_cuda, capi, orc, case, esn0, max_diff):
    """Two points of tests/test_demod_gpu.py::test_margin_sweep_cadu_identity on the twin (the demodulator half; the decoder behind it is
    the oracle's here): chunk-parallel soft symbols at an SNR where RS loses ~10 % / ~3 % of the frames still decode to the reference's list."""
    G.test_margin_sweep_cadu_identity(torch_cuda, capi, orc, case, esn0, max_diff, n=420)


def test_margin_sweep_below_loop_threshold(torch_cuda, capi, orc):
    G.test_margin_sweep_below_loop_threshold(torch_cuda, capi, orc, 2.5, n=420)
