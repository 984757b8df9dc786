"""tests/test_aos_gpu.py's cases against the HOST TWIN of aos_demux.hip (tests/emu): the CCSDS AOS step's kernels and its host state machine in the CPU suite."""
import numpy as np
import pytest

from oracle import pyref
from tests import test_aos_gpu as G
from tests.test_dvbs2_on_twin_cpu import capi  # noqa: F401  (fixture: the twin's binding)


def _np_helpers():
    def to_dev(a):
        a = np.ascontiguousarray(a)
        return (a, a.ctypes.data)

    def zeros_dev(n, dt):
        a = np.zeros(n, dtype=dt)
        return (a, a.ctypes.data)

    return to_dev, (lambda d: d[0]), zeros_dev


def test_vcdu_and_select_on_the_twin(capi):
    if not pyref.AosRef.available():
        pytest.skip("oracle/_ref/libsdref_aos.so not built")
    G.check_vcdu_and_select(capi, *_np_helpers())


@pytest.mark.parametrize("case", G.CASES, ids=[str(i) for i in range(len(G.CASES))])
def test_demux_on_the_twin(capi, case):
    if not pyref.AosRef.available():
        pytest.skip("oracle/_ref/libsdref_aos.so not built")
    G.check_demux(capi, *_np_helpers(), case)
