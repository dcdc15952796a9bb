import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "qcpinn-convection-diffusion-qiskit_amd"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub: str = ""):
    """Import the (hyphenated) product package or one of its submodules."""
    return importlib.import_module(PKG_NAME + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (these tests never fall back to CPU)")
    return torch.device("cuda", 0)
