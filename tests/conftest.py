import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref():
    from oracle import pyref
    if not pyref.ref_available():
        pytest.skip("oracle/_ref/libsdref.so not built (needs /root/reference at build time)")
    return pyref.ref()


@pytest.fixture(scope="session")
def port():
    from oracle import pyref
    if not pyref.port_available():
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return pyref.port()
