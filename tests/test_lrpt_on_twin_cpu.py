"""tests/test_lrpt_gpu.py's cases against the HOST TWIN of lrpt_decoder.hip (tests/emu): the Meteor LRPT decoder's kernels and host logic in the CPU suite."""
import pytest

from oracle import pyref
from tests import test_lrpt_gpu as G
from tests.test_aos_on_twin_cpu import _np_helpers
from tests.test_dvbs2_on_twin_cpu import capi  # noqa: F401  (fixture: the twin's binding)


TWIN = [0, 4, 5, 8, 9]  # (RS-marginal and pure-noise streams take a minute each on the emulated kernels: they stay with the GPU suite)


@pytest.mark.parametrize("case", [G.CASES[i] for i in TWIN], ids=[str(i) for i in TWIN])
def test_lrpt_decoder_on_the_twin(capi, case):
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_lrpt_decode")):
        pytest.skip("needs the compiled reference")
    G.check_decoder(capi, *_np_helpers(), case)


def test_lrpt_host_path_on_the_twin(capi):
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_lrpt_decode")):
        pytest.skip("needs the compiled reference")
    G.test_lrpt_host_path(capi)


@pytest.mark.parametrize("name,kw,mode", G.M2X_CASES)
def test_lrpt_m2x_interleaved_on_the_twin(capi, name, kw, mode):
    G.test_lrpt_m2x_interleaved(capi, name, kw, mode)
