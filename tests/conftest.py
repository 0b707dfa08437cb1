import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running oracle pinning (opt-in via ISING_SLOW=1)")


def _have_gpu() -> bool:
    try:
        import ising_gpu_amd as ig
        return ig.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _have_gpu():
        pytest.fail("no MI355X visible: -m gpu tests must run on the GPU box (there is no CPU fallback)")
    return True


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
