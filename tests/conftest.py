import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun); everything else is CPU-only")


@pytest.fixture(scope="session")
def swr():
    import swr_amd
    return swr_amd


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def gpu_ctx(swr):
    swr.build()
    ctx = swr.Context()
    yield ctx
    ctx.close()
