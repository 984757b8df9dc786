"""The GPU parity tests of the FEC engine (tests/test_fec_gpu.py) and the golden-fixture tests (tests/test_golden_gpu.py), the very same
test functions, collected a second time against the HOST TWIN (tests/emu): the unchanged fec_kernels.hip / fec_engine.hip on the
stand-in runtime, where the 64 fibers of a wave meet at every ballot / shuffle / DPP move. Host logic (lock search FSM, speculation
certificates, deframer walk, batching) and integer arithmetic are what this proves on a machine without a GPU; the GPU build itself
stays with -m gpu."""
import importlib.util
import os

import pytest

from oracle import pyref
from tests import test_fec_gpu as F
from tests import test_golden_gpu as GG
from tests import test_zz_punctured_gpu as PG
from tests.emu import build as emu_build
from tests.emu import fake_torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    return fake_torch


@pytest.fixture(scope="module")
def capi():
    if not os.path.exists(emu_build.CLANG):
        pytest.skip("no host clang++ to build the twin with")
    lib = emu_build.build()
    spec = importlib.util.spec_from_file_location("capi_host_twin3", os.path.join(ROOT, "satdump_amd", "capi.py"))
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
    assert m.LIB_PATH == lib
    return m


@pytest.fixture(scope="module")
def orc():
    return pyref.best()


for _mod, _pfx in ((F, ""), (GG, ""), (PG, "")):
    for _name in dir(_mod):
        if _name.startswith("test_") and _name not in globals():
            globals()[_name] = getattr(_mod, _name)
