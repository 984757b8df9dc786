"""The float math the DVB-S2 frame PLL kernel restates (glibc 2.35's atan2f / atanf, satdump_amd/csrc/dvbs2_demap.hip) against the host libm, in the CPU
suite: the kernel's own copy, compiled for the host by the twin (tests/emu), on 1e7 arguments -- arbitrary bit patterns, signal-like values, tiny and
huge ratios -- must give libm's bits. tools/check_atan2f.c is the same comparison as a stand-alone C program (6e7 arguments, bad=0)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.test_dvbs2_on_twin_cpu import capi  # noqa: F401  (fixture: the twin's binding)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_atan2f_equals_libm_on_1e7_arguments(capi):
    from tests.test_dvbs2_gpu import _dvbs2_ref_lib
    ref = _dvbs2_ref_lib()  # its sdref_atan2f loops over the host libm's atan2f
    rng = np.random.default_rng(99)
    bad = 0
    for part in range(5):
        n = 2_000_000
        if part % 2 == 0:
            y = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
            x = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
        else:
            y = (rng.standard_normal(n) * 10.0 ** rng.uniform(-8, 2, n)).astype(np.float32)
            x = (rng.standard_normal(n) * 10.0 ** rng.uniform(-8, 2, n)).astype(np.float32)
            x[::17] = 1.0   # the x == 1 shortcut
            y[::19] = 0.0
            x[::23] = 0.0
        ok = ~(np.isnan(x) | np.isnan(y))
        want = np.zeros(n, dtype=np.float32)
        ref.sdref_atan2f(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), n, want.ctypes.data_as(C.c_void_p))
        got = np.zeros(n, dtype=np.float32)
        assert capi.lib().sdhip_op_atan2f(0, y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), n, got.ctypes.data_as(C.c_void_p)) == 0
        bad += int(np.count_nonzero(got[ok].view(np.uint32) != want[ok].view(np.uint32)))
    assert bad == 0


def test_standalone_c_check_builds_and_passes_on_a_sample():
    """tools/check_atan2f.c compiles and reports bad=0 (its full 6e7-argument run takes ~10 s: run here as it is)."""
    exe = "/tmp/sdhip_check_atan2f"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "check_atan2f.c"), "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300).stdout
    assert "bad=0" in out, out
