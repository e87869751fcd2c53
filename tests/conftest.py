import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def hip_ctx():
    """One context shared by the GPU tests; sized for the largest in-test block."""
    from bwtc_amd import hip
    ctx = hip.Context(device=0, max_block_size=(64 << 20) + 1024)
    yield ctx
    ctx.close()
