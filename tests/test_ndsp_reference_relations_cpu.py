"""What DESIGN.md 4a says about the reference's TWO block families, checked on the reference itself (no product code involved): the ndsp blocks
(src-core/dsp/**, run on their own threads through DSPStream FIFOs by oracle/ref_wrap_ndsp.cpp) against the legacy blocks (src-core/common/dsp/**,
oracle/ref_wrap.cpp) on the same samples. These relations are why the ndsp chain runs on the legacy engine's kernels."""
import numpy as np
import pytest

from oracle import pyref
from tests.test_ndsp_gpu import _signal


@pytest.fixture(scope="module")
def both():
    if not (pyref.ref_available() and pyref.NdspRef.available()):
        pytest.skip("needs the compiled reference (oracle/_ref)")
    return pyref.ref(), pyref.NdspRef()


def test_agc_and_clock_recovery_compute_the_same_floats(both):
    leg, nd = both
    x = _signal("qpsk", 9000, esn0=9.0, seed=4)
    a = nd.run("agc_cc", {"rate": 1e-3, "reference": 0.6, "gain": 1.0, "max_gain": 65536.0}, x, buf=777)
    assert np.array_equal(a.view(np.uint32), leg.block(0, [1e-3, 0.6, 1.0, 65536.0], x).view(np.uint32))
    m = nd.run("clock_recovery_mm_cc", {"omega": 3.0}, x, buf=1000)
    assert np.array_equal(m.view(np.uint32), leg.block(3, [3.0, (8.7e-3) ** 2 / 4, 0.5, 8.7e-3, 0.005], x).view(np.uint32))


def test_fir_block_is_the_legacy_filter_minus_ntaps_outputs(both):
    """dsp/filter/fir.cpp:80-83 keeps ntaps samples back and starts its window one sample later: the legacy filter's stream without its first ntaps
    outputs (the aligned VOLK call only prepends zero taps: the same sums)."""
    leg, nd = both
    x = _signal("qpsk", 9000, esn0=9.0, seed=5)
    for alpha, ntaps in ((0.5, 31), (0.25, 21)):
        f = nd.run("rrc_fir_cc", {"samplerate": 6e6, "symbolrate": 2e6, "alpha": alpha, "ntaps": ntaps}, x, buf=1234)
        want = leg.block(1, [6e6, 2e6, alpha, ntaps], x)
        assert len(f) == len(want) - ntaps and np.array_equal(f.view(np.uint32), want[ntaps:].view(np.uint32))


def test_costas_blocks_differ_only_through_the_clip(both):
    """dsp/pll/costas.cpp:42 clips with branched_clip, common/dsp/pll/costas_loop.cpp:48 with branchless_clip (0.5 * (|x + 1| - |x - 1|) in float): other
    floats for most |e| < 1, so the two loops' outputs part after the first sample -- by rounding noise, not by design."""
    leg, nd = both
    x = _signal("qpsk", 9000, esn0=9.0, seed=6)
    c = nd.run("costas_cc", {"order": 4, "loop_bw": 0.004, "freq_limit": 1.0}, x)
    w = leg.block(2, [0.004, 4, 1.0], x)
    assert len(c) == len(w) and np.array_equal(c[:1].view(np.uint32), w[:1].view(np.uint32))
    assert not np.array_equal(c.view(np.uint32), w.view(np.uint32))
    assert np.max(np.abs(c - w)) < 1e-3 * np.max(np.abs(w))
