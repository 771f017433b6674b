import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    # the CPU oracle is test infrastructure; make sure its shared library exists
    from oracle import oracle
    oracle.build()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from deblurgs_amd import _lib
    _lib.lib()  # fails loudly if libdgs_hip.so is missing: GPU tests must run the native path
    return torch.device("cuda:0")
