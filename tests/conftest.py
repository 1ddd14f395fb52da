import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def ctx():
    """A gh_ctx on cuda:0 bound to torch's current stream.  GPU tests only."""
    import torch
    from gslam_amd import hip
    assert torch.cuda.is_available(), "GPU test collected on a box without a GPU"
    torch.cuda.set_device(0)
    c = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()
