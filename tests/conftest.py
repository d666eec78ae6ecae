import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "slow: multi-second CPU test")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return load


def _backend_params():
    return [pytest.param("emu", id="emu"), pytest.param("gpu", marks=pytest.mark.gpu, id="gpu")]


@pytest.fixture(scope="session", params=_backend_params())
def be(request):
    """Backend under test: 'emu' = the kernel sources on the host-side HIP emulator (tests/hipemu,
    CPU-only logic check); 'gpu' = the product libimgfd.so on cuda:0 through the C ABI."""
    import backends
    if request.param == "emu":
        return backends.EmuBackend()
    return backends.GpuBackend()
