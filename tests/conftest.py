import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def starfleet():
    with open(os.path.join(GOLDEN, "starfleet.html"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def compressor():
    """One sfh_ctx for the whole GPU session (gpu-marked tests only)."""
    from starflate_amd import Compressor

    c = Compressor(0)
    yield c
    c.close()
