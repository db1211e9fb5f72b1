import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


EXPERIMENT_KERNELS = (1, 3, 4, 5, 6, 9)   # attention families under csrc/experiments/ (not in the shipped library)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: exercises only the A/B kernels under csrc/experiments/ (collected with ICV_EXPERIMENTS=1)")
    config.addinivalue_line("markers", "experiment_kernels: parametrised over attention families; the experiments' cases are collected with ICV_EXPERIMENTS=1")


def pytest_collection_modifyitems(config, items):
    """The measured-slower A/B kernels are compiled only by `ICV_EXPERIMENTS=1 csrc/build.sh`; their tests are part of the
    collection only under the same switch (ICV_EXPERIMENTS=1 ICV_LIB_PATH=.../libicvideo_experiments.so pytest -m gpu),
    instead of showing up as dozens of skips in the default suite."""
    # the driver runs `pytest -x`: tests that need real multi-GPU RCCL ranks go LAST, so that a first-contact failure on a
    # multi-GPU node cannot leave the single-GPU rows (buffers, voxels, wire formats, ...) unrun
    last = ("test_multigpu_rccl.py",)
    items.sort(key=lambda it: 1 if os.path.basename(str(it.fspath)) in last else 0)      # stable: order inside each class kept
    if os.environ.get("ICV_EXPERIMENTS", "0") == "1":
        return
    keep, drop = [], []
    for it in items:
        exp = it.get_closest_marker("experiments") is not None
        if it.get_closest_marker("experiment_kernels") is not None and hasattr(it, "callspec"):
            exp = exp or it.callspec.params.get("kernel") in EXPERIMENT_KERNELS
        (drop if exp else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def hip_ops():
    """The product operator set on cuda:0.  Fails (does not skip) if the native library is missing:
    a GPU test must never pass on a fallback."""
    import torch
    from infinicube_amd.videogen.ops import HipOps
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return HipOps("cuda:0")
