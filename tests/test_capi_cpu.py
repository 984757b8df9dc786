"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/sdhip.h declares
(no compute calls: there is no GPU here and no CPU fallback), and the host-side config plumbing works."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sdhip.h")).read()
    return sorted(set(re.findall(r"\b(sdhip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from satdump_amd import capi
    L = capi.lib()
    missing = [s for s in _declared_symbols() if not hasattr(L, s)]
    assert not missing, f"declared in include/sdhip.h but not exported: {missing}"


def test_version_and_defaults():
    from satdump_amd import capi
    assert b"gfx950" in capi.lib().sdhip_version()
    c = capi.fec_cfg(constellation="qpsk", rs_i=4)
    assert c.constellation == capi.QPSK and c.asm_sync == 0x1ACFFC1D and c.rs_fill_bytes == -1 and c.derand_start == 4


def test_fails_loudly_without_gpu():
    import torch
    from satdump_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.SdhipError):
        capi.FecDecoder(capi.fec_cfg(constellation="bpsk"))
