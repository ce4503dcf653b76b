import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from backends import OracleBackend
    return OracleBackend()


@pytest.fixture(scope="session", autouse=True)
def torch_runtime_first(request):
    """GPU runs: bring PyTorch's HIP runtime up before libmpcx.so is loaded.  Some tests drive the library through
    torch tensors; when the library (linked against the system ROCm) has initialised HIP first, torch's own bundled
    runtime no longer finds the device ("No HIP GPUs are available")."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):  # deselected items are not listed
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    yield
