import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from _oracle import oracle_backend

    return oracle_backend()


@pytest.fixture(scope="session")
def hip():
    """The product backend: libpgbart_hip.so + torch device memory.  GPU tests only."""
    from pymc_bart_amd.sampler import default_backend

    be = default_backend()
    assert be.lib.backend_name == "hip-gfx950"
    return be
