import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    from stable_diffusion_training_amd import _lib, build
    build.build_library()
    return _lib.load()


@pytest.fixture(scope="session")
def dev(lib):
    import torch
    from stable_diffusion_training_amd import _lib
    _lib.require_device()
    assert torch.cuda.is_available(), "gpu tests need a visible HIP device"
    return torch.device("cuda:0")
