import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the oracle's stock-op convolutions (MIOpen) must not spend minutes auto-tuning dozens of one-off shapes on the GPU box
os.environ.setdefault("MIOPEN_FIND_MODE", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the test session is an entry point: on the GPU box it runs the configuration bench.py and the train mains run (kernel
    # arguments in device memory, the recorded library-GEMM selection) -- chosen here, before anything touches the GPU
    from emlight_amd import _runtime
    _runtime.entry_point_defaults()


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass without a GPU: they are skipped unless a
    device is visible, and on a GPU box a missing HIP library is a hard failure."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def case(self, prefix):
        p = prefix + "/"
        return {k[len(p):]: self.z[k] for k in self.z.files if k.startswith(p)}

    def __getitem__(self, k):
        return self.z[k]


@pytest.fixture(scope="session")
def golden_sinkhorn():
    return Golden("sinkhorn")


@pytest.fixture(scope="session")
def golden_raster():
    return Golden("rasteriser")


@pytest.fixture(scope="session")
def golden_densenet():
    return Golden("densenet")


@pytest.fixture(scope="session")
def golden_densenet_cfg2():
    return Golden("densenet_cfg2")
