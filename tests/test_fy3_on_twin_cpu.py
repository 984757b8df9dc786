"""tests/test_fy3_gpu.py's cases against the HOST TWIN of the FEC engine (tests/emu): the FengYun-3 AHRPT decoder's kernels and host logic in the CPU suite."""
import pytest

from oracle import pyref
from tests import test_fy3_gpu as G
from tests.test_aos_on_twin_cpu import _np_helpers
from tests.test_dvbs2_on_twin_cpu import capi  # noqa: F401  (fixture: the twin's binding)


import os

# (the emulated Viterbi kernels take tens of seconds per case: the rest stay with the GPU suite; FY3_TWIN_ALL=1 runs them all here, -n 8 helps)
TWIN = list(range(len(G.CASES))) if os.environ.get("FY3_TWIN_ALL") else [0]


@pytest.mark.parametrize("case", [G.CASES[i] for i in TWIN], ids=[str(i) for i in TWIN])
def test_fy3_decoder_on_the_twin(capi, case):
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_fy3_decode")):
        pytest.skip("needs the compiled reference")
    G.check_decoder(capi, *_np_helpers(), case)


@pytest.mark.skipif(not os.environ.get("FY3_TWIN_ALL"), reason="FY3_TWIN_ALL=1: the cut / host-path case on the twin")
def test_fy3_cuts_on_the_twin(capi):
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_fy3_decode")):
        pytest.skip("needs the compiled reference")
    G.check_cuts(capi, *_np_helpers())


@pytest.mark.parametrize("case", [G.MPT_CASES[0]] if not os.environ.get("FY3_TWIN_ALL") else G.MPT_CASES)
def test_fy3_mpt_decoder_on_the_twin(capi, case):
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_fy3_mpt_decode")):
        pytest.skip("needs the compiled reference")
    G.check_decoder(capi, *_np_helpers(), case)
