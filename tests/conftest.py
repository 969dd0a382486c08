import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure; built with gcc on first use)"""
    from oracle import pyorc
    pyorc.lib()
    return pyorc


@pytest.fixture(scope="session")
def hip():
    """the product library; GPU tests fail loudly if it is missing or no device is visible"""
    import piqp_amd
    L = piqp_amd._lib.load()
    assert L.pq_device_count() > 0, "no HIP device visible"
    return piqp_amd
