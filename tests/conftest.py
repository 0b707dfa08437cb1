import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    # the suite must exercise the in-tree product library, not an A/B build picked up through ISING_LIB
    assert "ISING_LIB" not in os.environ, "unset ISING_LIB: the tests run against ising_gpu_amd/libising_hip.so"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running oracle pinning (opt-in via ISING_SLOW=1)")


def _have_gpu() -> bool:
    import ising_gpu_amd as ig  # an import or load failure is an error of its own, not "no GPU"
    assert os.path.dirname(ig.LIB_PATH) == os.path.join(ROOT, "ising_gpu_amd")
    return ig.device_count() > 0


@pytest.fixture(scope="session")
def gpu():
    if not _have_gpu():
        pytest.fail("no MI355X visible: -m gpu tests must run on the GPU box (there is no CPU fallback)")
    return True


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    # the GPU box's host has 256 hardware threads shared with other tenants: an OpenMP team that wide spends its time in
    # barriers on the small lattices of the suite (a 5 s test becomes 0.3 s with 32 threads)
    oracle.set_threads(min(32, os.cpu_count() or 1))
    return oracle
