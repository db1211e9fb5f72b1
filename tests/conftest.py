import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ops():
    """The product operator set on cuda:0.  Fails (does not skip) if the native library is missing:
    a GPU test must never pass on a fallback."""
    import torch
    from infinicube_amd.videogen.ops import HipOps
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return HipOps("cuda:0")
